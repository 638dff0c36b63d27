// tools/ubench/copy_probe.hip -- MEASUREMENT TOOL, not product: how 133 MB device-to-host copies fare while persistent kernels hold every
// wavefront slot (the pixel stage's situation), by the way the copy is issued: hipMemcpyAsync on a stream of normal / high priority, on a
// stream with a CU mask (a hardware queue of its own), on two streams, a hand-written copy kernel, and the SDMA engines asked for directly
// through HSA (hsa_amd_memory_async_copy / _on_engine). Prints GB/s per way, idle and beside the hogs.
//   hipcc --offload-arch=gfx950 -O2 -o build/copy_probe tools/ubench/copy_probe.hip -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { fprintf(stderr, "%s:%d %s -> %d\n", __FILE__, __LINE__, #x, (int) s_); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// a stand-in for the pixel kernels: 256 lanes, 33 KB of LDS, every workgroup busy for `ticks` of the 100 MHz clock
__global__ void __launch_bounds__(256) k_hog(uint32_t *sink, uint64_t ticks) {
	__shared__ uint32_t lds[8448];
	for (int i = threadIdx.x; i < 8448; i += 256) lds[i] = i;
	__syncthreads();
	const uint64_t t0 = wall_clock64();
	uint32_t a = threadIdx.x;
	while (wall_clock64() - t0 < ticks) { for (int k = 0; k < 64; ++k) a = lds[(a * 13u + k) % 8448u] + a; }
	if (a == 0xdeadbeefu) sink[0] = a;
}

// the hand-written copy: `wgs` workgroups stride over the buffer, 16 bytes per lane per turn
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy(v4u *__restrict__ dst, const v4u *__restrict__ src, size_t n16) {
	for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) __builtin_nontemporal_store(src[i], &dst[i]);
}

static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_status_t agent_cb(hsa_agent_t a, void *) {
	hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
	if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
	if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
	return HSA_STATUS_SUCCESS;
}

int main(int argc, char **argv) {
	const size_t bytes = (size_t) 7680 * 4320 * 4;
	const int NB = 8, NCOPY = argc > 1 ? atoi(argv[1]) : 48;
	const int hog_launches = argc > 2 ? atoi(argv[2]) : 10;
	const unsigned way_mask = argc > 3 ? (unsigned) strtoul(argv[3], nullptr, 0) : 0xffffffffu;   // bit k: run way k
	const int only_beside = argc > 4 ? atoi(argv[4]) : -1;   // 0: idle only, 1: beside the hogs only
	CK(hipSetDevice(0));
	std::vector<void *> dev(NB), host(NB);
	for (int i = 0; i < NB; ++i) { CK(hipMalloc(&dev[i], bytes)); CK(hipMemset(dev[i], i + 1, bytes)); CK(hipHostMalloc(&host[i], bytes, hipHostMallocDefault)); memset(host[i], 0, bytes); }
	uint32_t *sink; CK(hipMalloc(&sink, 64));
	int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	printf("stream priorities: lowest %d, highest %d\n", lo, hi);
	hipStream_t hogs[4]; for (auto &s : hogs) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	hipStream_t s_norm, s_norm2, s_high, s_mask, s_mask2;
	CK(hipStreamCreateWithFlags(&s_norm, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_norm2, hipStreamNonBlocking));
	CK(hipStreamCreateWithPriority(&s_high, hipStreamNonBlocking, hi));
	hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
	const int cus = pr.multiProcessorCount; std::vector<uint32_t> mask((size_t) (cus + 31) / 32, 0xffffffffu);
	if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
	CK(hipExtStreamCreateWithCUMask(&s_mask, (uint32_t) mask.size(), mask.data())); CK(hipExtStreamCreateWithCUMask(&s_mask2, (uint32_t) mask.size(), mask.data()));
	printf("%d CUs; copies of %.1f MB, %d per test\n", cus, bytes / 1e6, NCOPY);
	// HSA
	HK(hsa_init()); HK(hsa_iterate_agents(agent_cb, nullptr));
	if (!have_gpu || !have_cpu) { fprintf(stderr, "no HSA agents\n"); return 1; }
	uint32_t engine_mask = 0;
	{ hsa_status_t s = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &engine_mask); printf("hsa_amd_memory_copy_engine_status(D2H): status %d, free engines mask 0x%x\n", (int) s, engine_mask); }
	std::vector<hsa_signal_t> sigs((size_t) NCOPY); for (auto &s : sigs) HK(hsa_signal_create(1, 0, nullptr, &s));

	auto start_hogs = [&](uint64_t ticks) { for (int k = 0; k < hog_launches; ++k) for (auto &s : hogs) hipLaunchKernelGGL(k_hog, dim3(32768), dim3(256), 0, s, sink, ticks); };
	auto wait_hogs = [&] { for (auto &s : hogs) CK(hipStreamSynchronize(s)); };
	struct Way { const char *name; int kind; };
	const Way ways[] = {{"hipMemcpyAsync, normal-priority stream", 0}, {"hipMemcpyAsync, high-priority stream", 1}, {"hipMemcpyAsync, CU-mask stream (all CUs)", 2},
		{"hipMemcpyAsync, two normal streams alternating", 3}, {"hipMemcpyAsync, two CU-mask streams alternating", 4}, {"copy kernel (64 workgroups), CU-mask stream", 5},
		{"copy kernel (256 workgroups), CU-mask stream", 6}, {"HSA hsa_amd_memory_async_copy (runtime picks the engine)", 7}, {"HSA ..._on_engine, one SDMA engine (lowest free), forced", 8},
		{"HSA ..._on_engine, two SDMA engines alternating, forced", 9}, {"hipMemcpyAsync, stream waits on a hog event first (as the pipeline's copy stream does)", 10}};
	if (way_mask >> 11 & 1u) {   // every SDMA engine by itself, both directions, idle device
		printf("---- each SDMA engine by itself (forced), idle device, 16 copies of %.1f MB\n", bytes / 1e6);
		uint32_t h2d_mask = 0; (void) hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &h2d_mask);
		printf("free engines: D2H 0x%x, H2D 0x%x\n", engine_mask, h2d_mask);
		for (int rep = 0; rep < 2; ++rep) for (int e = 0; e < 16; ++e) {
			double gbs[2] = {0, 0};
			for (int dir = 0; dir < 2; ++dir) {
				if (!((dir ? h2d_mask : engine_mask) >> e & 1u)) continue;
				const int n = 16; bool failed = false;
				const double t0 = now();
				for (int i = 0; i < n; ++i) {
					hsa_signal_store_relaxed(sigs[(size_t) i], 1);
					hsa_status_t st = dir ? hsa_amd_memory_async_copy_on_engine(dev[i % NB], g_gpu, host[i % NB], g_cpu, bytes, 0, nullptr, sigs[(size_t) i], (hsa_amd_sdma_engine_id_t) (1u << e), true)
					                      : hsa_amd_memory_async_copy_on_engine(host[i % NB], g_cpu, dev[i % NB], g_gpu, bytes, 0, nullptr, sigs[(size_t) i], (hsa_amd_sdma_engine_id_t) (1u << e), true);
					if (st != HSA_STATUS_SUCCESS) { failed = true; hsa_signal_store_relaxed(sigs[(size_t) i], 0); }
				}
				for (int i = 0; i < n; ++i) while (hsa_signal_wait_scacquire(sigs[(size_t) i], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {}
				gbs[dir] = failed ? -1.0 : n * (double) bytes / (now() - t0) / 1e9;
			}
			printf("pass %d engine %2d: D2H %6.1f GB/s   H2D %6.1f GB/s\n", rep, e, gbs[0], gbs[1]);
		}
		for (int i = 0; i < NB; ++i) CK(hipMemset(dev[i], i + 1, bytes));
		CK(hipDeviceSynchronize());
	}
	for (int beside = 0; beside < 2; ++beside) {
		printf("---- %s\n", beside ? "beside four streams of hog kernels (every wavefront slot taken)" : "idle device");
		if (only_beside >= 0 && only_beside != beside) continue;
		for (const Way &w : ways) {
			if (!(way_mask >> w.kind & 1u)) continue;
			if (w.kind >= 8 && engine_mask == 0) { printf("%-90s skipped (no engine reported free)\n", w.name); continue; }
			CK(hipDeviceSynchronize());
			if (beside) start_hogs(50000);   // 0.5 ms per workgroup, 16 rounds of 2 048 resident workgroups per launch and stream
			hipEvent_t ev = nullptr;
			if (w.kind == 10) { CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); hipLaunchKernelGGL(k_hog, dim3(256), dim3(256), 0, hogs[0], sink, (uint64_t) 1000); CK(hipEventRecord(ev, hogs[0])); CK(hipStreamWaitEvent(s_norm, ev, 0)); }
			const double t0 = now();
			int e0 = __builtin_ctz(engine_mask ? engine_mask : 1), e1 = e0;
			{ uint32_t rest = engine_mask & ~(1u << e0); if (rest) e1 = __builtin_ctz(rest); }
			for (int i = 0; i < NCOPY; ++i) {
				void *d = host[i % NB]; const void *s = dev[i % NB];
				switch (w.kind) {
					case 0: case 10: CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, s_norm)); break;
					case 1: CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, s_high)); break;
					case 2: CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, s_mask)); break;
					case 3: CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, i & 1 ? s_norm2 : s_norm)); break;
					case 4: CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToHost, i & 1 ? s_mask2 : s_mask)); break;
					case 5: hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s_mask, (v4u *) d, (const v4u *) s, bytes / 16); break;
					case 6: hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s_mask, (v4u *) d, (const v4u *) s, bytes / 16); break;
					case 7: hsa_signal_store_relaxed(sigs[(size_t) i], 1); HK(hsa_amd_memory_async_copy(d, g_cpu, s, g_gpu, bytes, 0, nullptr, sigs[(size_t) i])); break;
					case 8: hsa_signal_store_relaxed(sigs[(size_t) i], 1); HK(hsa_amd_memory_async_copy_on_engine(d, g_cpu, s, g_gpu, bytes, 0, nullptr, sigs[(size_t) i], (hsa_amd_sdma_engine_id_t) (1u << e0), true)); break;
					case 9: hsa_signal_store_relaxed(sigs[(size_t) i], 1); HK(hsa_amd_memory_async_copy_on_engine(d, g_cpu, s, g_gpu, bytes, 0, nullptr, sigs[(size_t) i], (hsa_amd_sdma_engine_id_t) (1u << (i & 1 ? e1 : e0)), true)); break;
				}
			}
			if (w.kind >= 7 && w.kind <= 9) { for (auto &s : sigs) while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {} }
			else { for (hipStream_t s : {s_norm, s_norm2, s_high, s_mask, s_mask2}) CK(hipStreamSynchronize(s)); }
			const double t1 = now();
			if (beside) wait_hogs();
			const double t2 = now();
			bool ok = true; for (int i = 0; i < NB && i < NCOPY; ++i) ok = ok && ((const uint8_t *) host[i])[12345] == (uint8_t) (i + 1) && ((const uint8_t *) host[i])[bytes - 1] == (uint8_t) (i + 1);
			for (int i = 0; i < NB; ++i) memset(host[i], 0, 4096), ((uint8_t *) host[i])[12345] = 0, ((uint8_t *) host[i])[bytes - 1] = 0;
			printf("%-90s %6.1f GB/s  (%.0f ms%s)%s\n", w.name, NCOPY * (double) bytes / (t1 - t0) / 1e9, (t1 - t0) * 1e3, beside ? (t2 - t1 > 1e-3 ? "; the hogs ran on" : "; the hogs had ENDED before the copies: lengthen them") : "", ok ? "" : "  WRONG BYTES");
			if (ev) CK(hipEventDestroy(ev));
		}
	}
	return 0;
}
