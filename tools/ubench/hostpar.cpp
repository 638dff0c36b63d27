// tools/ubench/hostpar.cpp -- host-side scaling probe: T threads each parse (+ plan-build) the same codestream K times.
// usage: hostpar file.jxl T K [plan]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../j40_amd/csrc/plan_build.hpp"
using namespace j40hip;
int main(int argc, char **argv) {
	FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); size_t n = (size_t) ftell(f); fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> d(n); if (fread(d.data(), 1, n, f) != n) return 1; fclose(f);
	const int T = atoi(argv[2]), K = atoi(argv[3]); const bool plan = argc > 4;
	auto t0 = std::chrono::steady_clock::now();
	std::vector<std::thread> th;
	std::vector<double> ms((size_t) T, 0.0);
	for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
		for (int k = 0; k < K; ++k) {
			auto a = std::chrono::steady_clock::now();
			Frame fr; const uint8_t *cs; size_t cs_size; std::vector<uint8_t> storage;
			extract_codestream(d.data(), n, &cs, &cs_size, &storage);
			parse_frame(cs, cs_size, &fr, 1);
			if (plan) { HostPlan hp; build_vardct_plan(fr, cs, cs_size, &hp); }
			ms[(size_t) t] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
		}
	});
	for (auto &x : th) x.join();
	const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	double sum = 0; for (double v : ms) sum += v;
	printf("threads %d: %d frames in %.3f s = %.1f frames/s, %.1f ms per frame per thread\n", T, T * K, wall, T * K / wall, sum / (T * K));
}
