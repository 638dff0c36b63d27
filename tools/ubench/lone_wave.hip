// tools/ubench/lone_wave.hip -- what ONE wavefront alone on its SIMD pays per instruction on gfx950 (MEASUREMENT TOOL, not product):
// dependent and independent chains of the instruction kinds the serial decoders (k_lf_rows, k_hf_entropy_fast, k_modular_*) are made
// of. Prints cycles per instruction (s_memtime) for 1 and for 2 wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O2 -o lone_wave lone_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define ITER 256

template <int KIND>
__global__ void __launch_bounds__(256) k(uint64_t *out, const uint32_t *chase, uint32_t *sink) {
	__shared__ uint32_t lds[2048];
	for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (uint32_t) (((i * 4u + 52u) & 8191u));   // byte offset of the next element (stride 13 dwords)
	__syncthreads();
	uint32_t a = threadIdx.x, b = threadIdx.x + 1, c = threadIdx.x + 2, d = threadIdx.x + 3, kk = sink[0] | 1u;
	uint64_t q = ((uint64_t) threadIdx.x << 32) | 0x12345u;
	uint32_t addr = (threadIdx.x & 63) * 4u;
	const uint32_t *gp = chase + (threadIdx.x & 63);
	const uint64_t t0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITER; ++it) {
		if (KIND == 0) { R16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(kk));) }
		if (KIND == 1) { R16(asm volatile("v_add_u32 %0, %0, %2\n v_add_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(kk));) }
		if (KIND == 2) { R16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk));) }
		if (KIND == 3) { R16(asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q));) }
		if (KIND == 4) { R16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));) }
		if (KIND == 5) { R16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(kk) : "vcc");) }
		if (KIND == 6) { R16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(kk));) }
		if (KIND == 7) { R16(asm volatile("v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_mov_b32 %0, s20" : "+v"(a) : : "s20", "scc");) }
		if (KIND == 8) { R16(asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");) }
		if (KIND == 9) { R16(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 1\n 1:\n s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(0xfffffff0u) : "vcc", "s20", "s21", "scc");) }   // branch never skipped... (exec non-zero)
		if (KIND == 10) { R16(asm volatile("v_cmp_gt_u32 vcc, %1, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, 1\n 1:\n s_or_b64 exec, exec, s[20:21]" : "+v"(a) : "v"(0u) : "vcc", "s20", "s21", "scc");) }   // always skipped (exec zero -> taken branch)
		if (KIND == 11) { R16(gp = chase + *(volatile const uint32_t *) gp;) }   // dependent global loads (L2 / L1 hits)
		if (KIND == 12) { R16({ const uint64_t v = ((volatile uint64_t *) lds)[addr >> 3]; addr = (uint32_t) v & 8184u; }) }
		if (KIND == 13) { R16(asm volatile("v_bfe_u32 %0, %0, 0, 31\n v_lshlrev_b32 %0, 1, %0" : "+v"(a));) }
		if (KIND == 15) { R16(asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a) : "v"(kk));) }   // the 8-byte encoding of the same instruction
		if (KIND == 16) { R16(asm volatile("v_add_u32_e64 %0, %0, %4\n v_add_u32_e64 %1, %1, %4\n v_add_u32_e64 %2, %2, %4\n v_add_u32_e64 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk));) }
		if (KIND == 17) { R16(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }   // 4 + 4 bytes of literal
		if (KIND == 18) { R16(asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk));) }
		if (KIND == 19) { R16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(kk) : "s20", "s21", "s22", "s23", "scc");) }   // vector and scalar side by side
		if (KIND == 14) { R16(sink[64 + (threadIdx.x & 63)] = a; gp = chase + *(volatile const uint32_t *) gp;) }   // a store in flight before every dependent load
	}
	const uint64_t t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
	sink[1 + (threadIdx.x & 31)] = a + b + c + d + (uint32_t) q + (uint32_t) (q >> 32) + addr + (uint32_t) (gp - chase);
}

template <int KIND> double run(int threads, int per, uint64_t *out, uint32_t *chase, uint32_t *sink) {
	hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, chase, sink);
	hipDeviceSynchronize();
	hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, chase, sink);
	hipDeviceSynchronize();
	uint64_t c = 0;
	hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
	return (double) c / (ITER * 16.0 * per);
}

int main() {
	uint64_t *out; uint32_t *chase, *sink;
	hipMalloc(&out, 64); hipMalloc(&chase, 4096 * 4); hipMalloc(&sink, 4096);
	std::vector<uint32_t> h(4096);
	for (int i = 0; i < 4096; ++i) h[i] = (uint32_t) ((i + 64 * 17) & 4095);   // next = + 17 lines of 64 dwords
	hipMemcpy(chase, h.data(), 4096 * 4, hipMemcpyHostToDevice);
	hipMemset(sink, 0, 4096);
	int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	printf("{\"clock_khz\": %d,\n", clk); fflush(stdout);
	const char *names[] = {"v_add dependent", "v_add 2 chains (per instr)", "v_add 4 chains (per instr)", "v_lshrrev_b64 dependent", "ds_read_b32 chase", "v_cmp+v_cndmask pair",
		"v_mul_lo_u32 dependent", "readfirstlane+s_add+v_mov triple", "s_add dependent", "if-block not skipped (5 instr)", "if-block skipped, branch taken (4 instr)", "global load chase", "ds_read_b64 chase", "v_bfe+v_lshl pair", "store + global load chase", "v_add_u32_e64 (8-byte encoding) dependent", "v_add_u32_e64 4 chains (per instr)", "v_add_u32 with a 32-bit literal, 4 chains (per instr)", "v_add3_u32 4 chains (per instr)", "4 v_add chains + 4 s_add chains (per instr)"};
	for (int w = 0; w < 2; ++w) {   // (a third row of 512 lanes -- two wavefronts per SIMD -- never launched under __launch_bounds__(256) and repeated a stale counter: dropped)
		const int threads = w == 0 ? 64 : 256;   // 1 wavefront; 4 = one per SIMD
		printf(" \"%d wavefronts in the workgroup (cycles per unit on wavefront 0)\": {", threads / 64);
		typedef double (*RunFn)(int, int, uint64_t *, uint32_t *, uint32_t *);
		const RunFn fns[20] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>, run<13>, run<14>, run<15>, run<16>, run<17>, run<18>, run<19>};
		const int per[20] = {1, 2, 4, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 4, 4, 4, 8};
		for (int i = 0; i < 20; ++i) { printf("%s\"%s\": %.1f", i ? ", " : "", names[i], fns[i](threads, per[i], out, chase, sink)); fflush(stdout); }
		printf("}%s\n", w < 1 ? "," : "");
	}
	printf("}\n");
	return 0;
}
