// tools/jxlsynth.cpp -- synthetic JPEG XL stream generator (test/bench infrastructure).
//
// No encoder or sample file exists offline, so this tool writes the streams
// the tests and bench.py decode: VarDCT frames (all 27 transform types, rANS with clustered
// contexts, optional custom block-context map / coefficient orders / HF presets / multiple passes)
// and Modular frames (single- and multi-group, RCT / Palette, prefix codes with LZ77, weighted
// predictor). Two ways to make a VarDCT frame:
//   forward=1  an ENCODE of a procedural picture: XYB forward, LF = 8x8 means, forward transform per varblock
//              (jxlsynth_forward.hpp), quantisation at about distance 1 with the library matrices, transform sizes and
//              HfMul chosen from local activity. Size, symbols per pixel and the spread of the sections' lengths follow
//              from the picture (1/f detail whose strength varies over the frame: `detail=`). What bench.py decodes.
//   forward=0  synthesis in the *coefficient domain* (the feature-matrix streams of the tests and the golden fixtures): the
//              LF image comes from a smooth procedural picture, HF coefficients are sparse with frequency-decaying
//              magnitudes; every transform type and bitstream feature can be forced into a small frame.
// The reference decoder (oracle/_ref) is the judge of validity; what a stream decodes to is defined by it.
//
// usage: jxlsynth vardct  W H SEED OUT [key=value ...]
//        jxlsynth modular W H SEED OUT [key=value ...]
#include "jxlsynth_common.hpp"
#include "jxlsynth_modular.hpp"
#include "jxlsynth_forward.hpp"
#include <cmath>
#include <map>
#include <functional>

using namespace synth;

// ------------------------------------------------------------------------------------------------
// options

struct Options {
	std::map<std::string, std::string> kv;
	int geti(const char *k, int def) const { auto it = kv.find(k); return it == kv.end() ? def : atoi(it->second.c_str()); }
	double getd(const char *k, double def) const { auto it = kv.find(k); return it == kv.end() ? def : atof(it->second.c_str()); }
	std::string gets(const char *k, const char *def) const { auto it = kv.find(k); return it == kv.end() ? def : it->second; }
};

// ------------------------------------------------------------------------------------------------
// procedural source picture (deterministic, cheap): low-frequency sinusoids + value noise +
// hard-edged rectangles, sRGB in [0,1]

struct Picture {
	uint64_t seed;
	struct Rect { int x0, y0, x1, y1; float r, g, b; };
	std::vector<Rect> rects;
	float fx[6], fy[6], ph[6], amp[6][3];
	int W, H;
	Picture(int w, int h, uint64_t s) : seed(s), W(w), H(h) {
		SplitMix64 rng(s ^ 0x4A34304Aull);
		for (int i = 0; i < 6; ++i) {
			fx[i] = (float) (rng.unit() * 6.0 / w); fy[i] = (float) (rng.unit() * 6.0 / h); ph[i] = (float) (rng.unit() * 6.28318);
			for (int c = 0; c < 3; ++c) amp[i][c] = (float) (rng.unit() * 0.09);
		}
		int nrect = 24;
		for (int i = 0; i < nrect; ++i) {
			Rect r; r.x0 = (int) rng.below((uint32_t) w); r.y0 = (int) rng.below((uint32_t) h);
			r.x1 = r.x0 + 8 + (int) rng.below((uint32_t) std::max(9, w / 4)); r.y1 = r.y0 + 8 + (int) rng.below((uint32_t) std::max(9, h / 4));
			r.r = (float) (rng.unit() * 0.3 - 0.15); r.g = (float) (rng.unit() * 0.3 - 0.15); r.b = (float) (rng.unit() * 0.3 - 0.15);
			rects.push_back(r);
		}
	}
	static float hash01(uint64_t a) { a *= 0x9e3779b97f4a7c15ull; a ^= a >> 29; a *= 0xbf58476d1ce4e5b9ull; a ^= a >> 32; return (float) (a & 0xffffff) * (1.0f / 16777216.0f); }
	float vnoise(float x, float y, int oct, int ch) const {
		float sc = (float) (1 << oct) * 4.0f / (float) std::max(W, H);
		float u = x * sc, v = y * sc; int iu = (int) u, iv = (int) v; float fu = u - (float) iu, fv = v - (float) iv;
		auto hv = [&](int a, int b) { return hash01(seed * 31 + (uint64_t) a * 73856093ull + (uint64_t) b * 19349663ull + (uint64_t) oct * 83492791ull + (uint64_t) ch * 2654435761ull); };
		float su = fu * fu * (3 - 2 * fu), sv = fv * fv * (3 - 2 * fv);
		return (hv(iu, iv) * (1 - su) + hv(iu + 1, iv) * su) * (1 - sv) + (hv(iu, iv + 1) * (1 - su) + hv(iu + 1, iv + 1) * su) * sv;
	}
	// the smooth part of the picture (no rectangles, not clamped) / the rectangles' contribution: forward=1 evaluates the former
	// on a coarse grid
	void smooth(float x, float y, float out[3]) const {
		for (int c = 0; c < 3; ++c) {
			float v = 0.5f;
			for (int i = 0; i < 6; ++i) v += amp[i][c] * sinf(fx[i] * x * 6.28318f + fy[i] * y * 6.28318f + ph[i] + (float) c);
			for (int o = 0; o < 3; ++o) v += (vnoise(x, y, o, c) - 0.5f) * (0.16f / (float) (1 << o));
			out[c] = v;
		}
	}
	void add_rects(float x, float y, float out[3]) const {
		for (const Rect &r : rects) if (x >= (float) r.x0 && x < (float) r.x1 && y >= (float) r.y0 && y < (float) r.y1) { out[0] += r.r; out[1] += r.g; out[2] += r.b; }
	}
	void rgb(float x, float y, float out[3]) const {
		for (int c = 0; c < 3; ++c) {
			float v = 0.5f;
			for (int i = 0; i < 6; ++i) v += amp[i][c] * sinf(fx[i] * x * 6.28318f + fy[i] * y * 6.28318f + ph[i] + (float) c);
			for (int o = 0; o < 3; ++o) v += (vnoise(x, y, o, c) - 0.5f) * (0.16f / (float) (1 << o));
			out[c] = v;
		}
		for (const Rect &r : rects) if (x >= (float) r.x0 && x < (float) r.x1 && y >= (float) r.y0 && y < (float) r.y1) { out[0] += r.r; out[1] += r.g; out[2] += r.b; }
		for (int c = 0; c < 3; ++c) out[c] = std::min(0.92f, std::max(0.08f, out[c]));
	}
};

// ------------------------------------------------------------------------------------------------
// headers

static void write_size_header(BitWriter &bw, int w, int h) {  // inverse of j40.h:3008
	if (w % 8 == 0 && h % 8 == 0 && w <= 256 && h <= 256) {
		bw.put(1, 1); bw.put((uint64_t) (h / 8 - 1), 5);
		if (w == h) bw.put(1, 3); else { bw.put(0, 3); bw.put((uint64_t) (w / 8 - 1), 5); }
	} else {
		bw.put(0, 1); bw.u32(h, 1, 9, 1, 13, 1, 18, 1, 30);
		if (w == h) bw.put(1, 3); else { bw.put(0, 3); bw.u32(w, 1, 9, 1, 13, 1, 18, 1, 30); }
	}
}

static void write_toc_entry(BitWriter &bw, size_t size) { bw.u32((int64_t) size, 0, 10, 1024, 14, 17408, 22, 4211712, 30); }  // j40.h:5529

// IEEE half for values that are exactly representable (all the parameters below are)
static uint32_t f16_bits(float v) {
	if (v == 0.0f) return 0;
	uint32_t u; memcpy(&u, &v, 4);
	const uint32_t sign = u >> 31, mant = u & 0x7fffff; const int e = (int) ((u >> 23) & 0xff) - 127 + 15;
	if (e <= 0 || e >= 31 || (mant & 0x1fff)) die("f16: value not exactly representable");
	return sign << 15 | (uint32_t) e << 10 | mant >> 13;
}

// dq=1: the dequantisation matrices of HfGlobal in their coded forms (j40.h:4696-4760) instead of "all default": band
// parameters for DCT8, the Hornuss, DCT2x2, DCT4x4, DCT4x8 and AFV forms (the reference accepts the coded forms only for the
// 8x8 matrices, band parameters included: HOW[6].requires8x8, j40.h:4751); values near the library's, rounded to halves. Scaled parameters are stored divided by 64.
static void write_dq_matrices(BitWriter &bw, std::vector<std::pair<int, StreamEncoder>> &raw) {
	auto params = [&](const std::vector<std::array<float, 3>> &p, size_t first, size_t count, size_t scaled) {   // channel-major, as read (j40.h:4738)
		for (int c = 0; c < 3; ++c) for (size_t j = 0; j < count; ++j) bw.put(f16_bits(p[first + j][(size_t) c] / (j < scaled ? 64.0f : 1.0f)), 16);
	};
	auto bands = [&](const std::vector<std::array<float, 3>> &p) { bw.put((uint64_t) (p.size() - 1), 4); params(p, 0, p.size(), 1); };
	const std::vector<std::array<float, 3>> dct8 = {{3136, 576, 512}, {0, 0, -2}, {-0.5f, -0.25f, -1}, {-0.5f, -0.25f, 0}, {-0.5f, -0.25f, -1}, {-2, -0.25f, -2}};
	const std::vector<std::array<float, 3>> b4x4 = {{2176, 384, 112}, {0, 0, -0.25f}, {0, 0, -0.25f}, {0, 0, -0.5f}};
	const std::vector<std::array<float, 3>> b4x8 = {{2176, 768, 512}, {-1, -1, -1.5f}, {-0.75f, -1, -1.5f}, {-0.625f, -0.25f, -1.5f}};
	for (int idx = 0; idx < 17; ++idx) {
		switch (idx) {
		case 0: bw.put(6, 3); bands(dct8); break;
		case 1: { bw.put(1, 3); const std::vector<std::array<float, 3>> p = {{256, 64, 16}, {3200, 896, 192}, {3072, 832, 208}}; params(p, 0, 3, 3); break; }
		case 2: { bw.put(2, 3); const std::vector<std::array<float, 3>> p = {{3840, 960, 640}, {2560, 640, 320}, {1280, 320, 128}, {640, 180, 64}, {480, 140, 32}, {300, 120, 16}}; params(p, 0, 6, 6); break; }
		case 3: { bw.put(3, 3); const std::vector<std::array<float, 3>> p = {{1, 1, 1}, {2, 1, 0.5f}}; params(p, 0, 2, 2); bands(b4x4); break; }
		case 9: { bw.put(4, 3); const std::vector<std::array<float, 3>> p = {{1, 0.75f, 1.5f}}; params(p, 0, 1, 0); bands(b4x8); break; }
		case 10: {
			bw.put(5, 3);
			const std::vector<std::array<float, 3>> p = {{3072, 1024, 384}, {3072, 1024, 384}, {256, 48, 12}, {256, 48, 12}, {256, 48, 12}, {416, 56, 22}, {0, 0, -0.25f}, {0, 0, -0.25f}, {0, 0, -0.25f}};
			params(p, 0, 9, 6); bands(b4x8); bands(b4x4);
			break;
		}
		default: {
			StreamEncoder *enc = nullptr;
			for (auto &r : raw) if (r.first == idx) enc = &r.second;
			if (!enc) { bw.put(0, 3); break; }   // library
			// raw (j40.h:4712-4745): a denominator, then the weights times it as a three-channel Modular image
			bw.put(7, 3); bw.put(f16_bits(2.0f), 16);
			write_modular_header(bw, true, nullptr, {});
			enc->flush(bw);
		} }
	}
}

// BitDepth of ImageMetadata (j40.h:3175-3190): integer samples, U32(8, 10, 12, 1 + u(6)) bits
static void write_bit_depth(BitWriter &cs, int bpp) {
	cs.put(0, 1);   // not float
	if (bpp == 8) cs.put(0, 2); else if (bpp == 10) cs.put(1, 2); else if (bpp == 12) cs.put(2, 2); else { cs.put(3, 2); cs.put((uint64_t) (bpp - 1), 6); }
}

// TOC + sections (j40.h:5505-5543). permute != 0: the sections are stored in a shuffled order and the TOC carries the
// Lehmer-coded permutation that puts them back (the decoder applies it to the list of stored sections, j40.h:5540).
static void write_toc_and_sections(BitWriter &cs, const std::vector<std::vector<uint8_t>> &sections_in, int permute, SplitMix64 &rng, int slack = 0) {
	// slack=K: K junk bytes behind the data of every non-empty section. The reference does not notice them in frames with more
	// than one section (j40__finish_section_state drops the error of its own j40__no_more_bytes, j40.h:7778-7795)
	std::vector<std::vector<uint8_t>> sections = sections_in;
	if (slack) for (auto &s : sections) if (!s.empty()) for (int k = 0; k < slack; ++k) s.push_back((uint8_t) (0xa5 + 17 * k));
	const size_t n = sections.size();
	std::vector<size_t> stored_at(n);   // logical section i is the stored_at[i]-th stored one
	for (size_t i = 0; i < n; ++i) stored_at[i] = i;
	if (!permute || n < 3) cs.put(0, 1);
	else {
		const size_t end = n - 1 - (size_t) rng.below((uint32_t) std::min<size_t>(n - 2, 5));
		std::vector<uint32_t> lehmer(end);
		for (size_t i = 0; i < end; ++i) lehmer[i] = rng.below((uint32_t) std::min<size_t>(n - i, 40));
		{   // what the decoder's j40__apply_permutation makes of the stored list
			size_t *target = stored_at.data();
			for (uint32_t x : lehmer) { size_t tmp = target[x]; memmove(target + 1, target, sizeof(size_t) * x); target[0] = tmp; ++target; }
		}
		cs.put(1, 1);
		CodeSpecW pspec; pspec.init(8, std::vector<uint8_t>(8, 0), 1); pspec.log_alpha = 6; pspec.cfg[0] = HybridCfg{4, 1, 0};
		StreamEncoder penc(pspec);
		penc.add((uint32_t) std::min(7, ceil_lg((uint32_t) n + 1)), (uint32_t) end);   // j40.h:5437
		uint32_t prev = 0;
		for (uint32_t x : lehmer) { penc.add((uint32_t) std::min(7, ceil_lg(prev + 1)), x); prev = x; }
		count_stream(pspec, penc);
		write_code_spec(cs, pspec); penc.flush(cs);
	}
	cs.pad();
	std::vector<const std::vector<uint8_t> *> stored(n, nullptr);
	for (size_t i = 0; i < n; ++i) stored[stored_at[i]] = &sections[i];
	for (const auto *s : stored) write_toc_entry(cs, s->size());
	cs.pad();
	for (const auto *s : stored) cs.append_bytes(*s);
}

// transform table: the decoder's view (J40__DCT_SELECT, j40.h:4591): log rows, log columns, order
static const int8_t DCTSEL[27][3] = {
	{3,3,0},{3,3,1},{3,3,1},{3,3,1},{4,4,2},{5,5,3},{4,3,4},{3,4,4},{5,3,5},{3,5,5},{5,4,6},{4,5,6},{3,3,1},{3,3,1},
	{3,3,1},{3,3,1},{3,3,1},{3,3,1},{6,6,7},{6,5,8},{5,6,8},{7,7,9},{7,6,10},{6,7,10},{8,8,11},{8,7,12},{7,8,12},
};

// coefficient-context tables (spec constants as used at j40.h:6935-6947), pre-doubled
static const int8_t FREQ_CTX2[64] = {
	-1, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 30, 32, 32, 34, 34, 36, 36, 38, 38, 40, 40, 42, 42, 44, 44,
	46, 46, 46, 46, 48, 48, 48, 48, 50, 50, 50, 50, 52, 52, 52, 52, 54, 54, 54, 54, 56, 56, 56, 56, 58, 58, 58, 58, 60, 60, 60, 60,
};
static int nnz_ctx2(int q) {
	static const int16_t LIMIT[8] = {2, 3, 5, 9, 13, 21, 33, 64}, VALUE[8] = {0, 62, 124, 186, 246, 304, 360, 412};
	for (int i = 0; i < 8; ++i) if (q < LIMIT[i]) return VALUE[i];
	return 412;
}

struct LfGroupW {
	int left, top, w, h, w8, h8, w64, h64;
	std::vector<int32_t> blocks;               // per cell: (dctsel + 2) << 20 | voff, 1 << 20 | voff, or 0
	struct VB { int x8, y8, dctsel, hfmul_m1; };
	std::vector<VB> vbs;
	Channel lfq[3];                            // Y, X, B order as streamed
	Channel xfromy, bfromy, blockinfo, sharp;
	std::vector<uint8_t> lfidx;
};

static int run_vardct(int W, int H, uint64_t seed, const char *out, const Options &opt) {
	SplitMix64 rng(seed * 0x100000001b3ull + 12345);
	const int global_scale = opt.geti("global_scale", 8192), quant_lf = opt.geti("quant_lf", 4);
	const double density = opt.getd("density", 0.80), decay = opt.getd("decay", 0.84);
	const int max_log = opt.geti("maxlog", 6);              // largest transform side (log2) used in the mix
	const int coverage = opt.geti("coverage", opt.geti("forward", 0) ? 0 : 1);           // 1: force every transform type <= maxlog at least once per frame
	const int custom_bctx = opt.geti("bctx", 0);            // custom block context map with LF/QF thresholds
	const int num_presets = opt.geti("presets", 1);
	const int custom_orders = opt.geti("orders", 0);
	const int num_passes = opt.geti("passes", 1);
	const int nonzero_header = opt.geti("fullheader", (num_passes > 1) ? 1 : 0);
	const int x_qm = opt.geti("xqm", 3), b_qm = opt.geti("bqm", 2);
	const int skip_smooth = opt.geti("nosmooth", 0);
	const int small_clusters = opt.geti("simpleclusters", 0); // <= 8 clusters: simple cluster-map form
	const int log_alpha = opt.geti("logalpha", 7);
	const int container = opt.geti("container", 0);

	const int gcols = (W + 255) / 256, grows = (H + 255) / 256, num_groups = gcols * grows;
	const int ggcols = (W + 2047) / 2048, ggrows = (H + 2047) / 2048, num_lf_groups = ggcols * ggrows;
	if (num_groups < 2 && num_passes == 1) die("vardct: need at least 2 groups (single-section VarDCT order quirk, SURVEY section 0 fact 8)");

	Picture pic(W, H, seed);

	// ---- forward opsin (numerical inverse of the decoder's default inverse matrix, j40.h:3109) ----
	const double inv[3][3] = {
		{11.031566901960783, -9.866943921568629, -0.16462299647058826},
		{-3.254147380392157, 4.418770392156863, -0.16462299647058826},
		{-3.6588512862745097, 2.7129230470588235, 1.9459282392156863}};
	double fwd[3][3];
	{
		double a = inv[0][0], b = inv[0][1], c = inv[0][2], d = inv[1][0], e = inv[1][1], f = inv[1][2], g = inv[2][0], h = inv[2][1], i = inv[2][2];
		double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
		fwd[0][0] = (e * i - f * h) / det; fwd[0][1] = (c * h - b * i) / det; fwd[0][2] = (b * f - c * e) / det;
		fwd[1][0] = (f * g - d * i) / det; fwd[1][1] = (a * i - c * g) / det; fwd[1][2] = (c * d - a * f) / det;
		fwd[2][0] = (d * h - e * g) / det; fwd[2][1] = (b * g - a * h) / det; fwd[2][2] = (a * e - b * d) / det;
	}
	const double bias = -0.0037930732552754493, cbias = cbrt(bias);
	auto to_xyb = [&](const float rgb[3], double xyb[3]) {
		double lin[3];
		for (int c = 0; c < 3; ++c) { double v = rgb[c]; lin[c] = v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4); }
		double mix[3];
		for (int c = 0; c < 3; ++c) mix[c] = cbrt(fwd[c][0] * lin[0] + fwd[c][1] * lin[1] + fwd[c][2] * lin[2] - bias) + cbias;
		xyb[0] = (mix[0] - mix[1]) * 0.5; xyb[1] = (mix[0] + mix[1]) * 0.5; xyb[2] = mix[2];
	};
	// LF dequant steps (j40.h:6562): m_lf_scaled / (global_scale * quant_lf) * 65536
	const double m_lf[3] = {1.0 / 4096, 1.0 / 512, 1.0 / 256};
	double lfstep[3]; for (int c = 0; c < 3; ++c) lfstep[c] = m_lf[c] / ((double) global_scale * quant_lf) * 65536.0;

	// ---- forward=1: the picture's XYB samples (planes padded to whole cells, edges replicated) ----
	const int forward = opt.geti("forward", 0);
	if (forward && (num_passes > 1 || max_log > 6 || opt.geti("cfl", 0))) die("vardct: forward=1 takes one pass, transforms up to 64x64, default chroma-from-luma");
	const int PW = (W + 7) / 8 * 8, PH = (H + 7) / 8 * 8;
	std::vector<float> plane[3];
	std::unique_ptr<synthfwd::Forward> fwdx;
	if (forward) {
		fwdx.reset(new synthfwd::Forward());
		// 1/f detail: octaves of lattice noise from 256-pixel cells down to 2-pixel cells, amplitude falling with the cell size,
		// its strength modulated by a slowly varying mask (calm and busy regions, like sky and foliage)
		const double detail = opt.getd("detail", 2.4), beta = opt.getd("beta", 0.35);
		struct Lattice { int cell, w, h; float amp; std::vector<float> v; };
		std::vector<Lattice> lat;
		SplitMix64 lr(seed ^ 0xD37A11ull);
		for (int cell = 256; cell >= 2; cell /= 2) {
			Lattice l; l.cell = cell; l.w = PW / cell + 2; l.h = PH / cell + 2; l.amp = (float) (0.2 * detail * pow((double) cell / 256.0, beta));
			l.v.resize((size_t) l.w * (size_t) l.h);
			for (auto &v : l.v) v = (float) lr.unit() - 0.5f;
			lat.push_back(std::move(l));
		}
		Lattice mask; mask.cell = 512; mask.w = PW / 512 + 2; mask.h = PH / 512 + 2; mask.amp = 1.0f; mask.v.resize((size_t) mask.w * (size_t) mask.h);
		for (auto &v : mask.v) { const double u = lr.unit(); v = (float) (0.12 + 1.5 * u * u); }
		auto sample = [](const Lattice &l, int x, int y) {
			const float u = (float) x / (float) l.cell, v = (float) y / (float) l.cell;
			const int iu = (int) u, iv = (int) v; const float fu = u - (float) iu, fv = v - (float) iv;
			const float *p = l.v.data() + (size_t) iv * (size_t) l.w + (size_t) iu;
			return (p[0] * (1 - fu) + p[1] * fu) * (1 - fv) + (p[l.w] * (1 - fu) + p[l.w + 1] * fu) * fv;
		};
		for (int c = 0; c < 3; ++c) plane[c].resize((size_t) PW * (size_t) PH);
		const std::string dumpsrc = opt.gets("dumpsrc", "");   // the source picture as sRGB u8 x 3, for the fidelity test
		std::vector<uint8_t> srcdump(dumpsrc.empty() ? 0 : (size_t) W * (size_t) H * 3);
		static const float CHROMA[3] = {1.0f, 0.92f, 0.8f};
		// the smooth part of the picture on a grid of 4-pixel cells (it has no feature below ~100 pixels), sRGB -> linear by table
		const int GW = PW / 4 + 2, GH = PH / 4 + 2;
		std::vector<float> coarse((size_t) GW * (size_t) GH * 3);
		for (int gy = 0; gy < GH; ++gy) for (int gx = 0; gx < GW; ++gx) pic.smooth((float) (gx * 4), (float) (gy * 4), &coarse[((size_t) gy * (size_t) GW + (size_t) gx) * 3]);
		std::vector<float> to_linear(4097);
		for (int i = 0; i <= 4096; ++i) { const double v = i / 4096.0; to_linear[(size_t) i] = (float) (v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4)); }
		for (int y = 0; y < PH; ++y) for (int x = 0; x < PW; ++x) {
			const int sx = std::min(x, W - 1), sy = std::min(y, H - 1);
			float rgb[3]; double xyb[3];
			{
				const int gx = sx >> 2, gy = sy >> 2; const float fu = (float) (sx & 3) * 0.25f, fv = (float) (sy & 3) * 0.25f;
				const float *p = &coarse[((size_t) gy * (size_t) GW + (size_t) gx) * 3];
				for (int c = 0; c < 3; ++c) rgb[c] = (p[c] * (1 - fu) + p[3 + c] * fu) * (1 - fv) + (p[(size_t) GW * 3 + c] * (1 - fu) + p[(size_t) GW * 3 + 3 + c] * fu) * fv;
				pic.add_rects((float) sx, (float) sy, rgb);
			}
			float n = 0;
			for (const Lattice &l : lat) n += l.amp * sample(l, sx, sy);
			n *= sample(mask, sx, sy);
			for (int c = 0; c < 3; ++c) rgb[c] = std::min(0.97f, std::max(0.03f, rgb[c] + n * CHROMA[c]));
			if (!dumpsrc.empty() && x < W && y < H) for (int c = 0; c < 3; ++c) srcdump[((size_t) y * (size_t) W + (size_t) x) * 3 + (size_t) c] = (uint8_t) lrintf(rgb[c] * 255.0f);
			{   // to_xyb with the table for the transfer function (quantise the sample to the table's grid first: what is
				// dumped as the source is an 8-bit picture anyway)
				double lin[3], mix[3];
				for (int c = 0; c < 3; ++c) { const float t = rgb[c] * 4096.0f; const int i = (int) t; const float f = t - (float) i; lin[c] = to_linear[(size_t) i] * (1 - f) + to_linear[(size_t) std::min(i + 1, 4096)] * f; }
				for (int c = 0; c < 3; ++c) mix[c] = cbrt(fwd[c][0] * lin[0] + fwd[c][1] * lin[1] + fwd[c][2] * lin[2] - bias) + cbias;
				xyb[0] = (mix[0] - mix[1]) * 0.5; xyb[1] = (mix[0] + mix[1]) * 0.5; xyb[2] = mix[2];
			}
			for (int c = 0; c < 3; ++c) plane[c][(size_t) y * (size_t) PW + (size_t) x] = (float) xyb[c];
		}
		if (!dumpsrc.empty()) { FILE *fp = fopen(dumpsrc.c_str(), "wb"); if (!fp || fwrite(srcdump.data(), 1, srcdump.size(), fp) != srcdump.size()) die("cannot write dumpsrc"); fclose(fp); }
	}
	// activity of a cell: variance of its Y samples
	std::vector<float> activity(forward ? (size_t) (PW / 8) * (size_t) (PH / 8) : 0);
	for (int cy = 0; forward && cy < PH / 8; ++cy) for (int cx = 0; cx < PW / 8; ++cx) {
		double s = 0, s2 = 0;
		for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) { const double v = plane[1][(size_t) (cy * 8 + y) * (size_t) PW + (size_t) (cx * 8 + x)]; s += v; s2 += v * v; }
		activity[(size_t) cy * (size_t) (PW / 8) + (size_t) cx] = (float) std::max(0.0, s2 / 64 - (s / 64) * (s / 64));
	}
	auto cell_activity = [&](int cx, int cy) { return (double) activity[(size_t) cy * (size_t) (PW / 8) + (size_t) cx]; };

	const int custom_cfl = opt.geti("cfl", 0);
	const double kx_lf = custom_cfl ? 0.125 + 3.0 / 128.0 : 0.0, kb_lf = custom_cfl ? 0.75 - 2.0 / 128.0 : 1.0;

	// ---- block context configuration ----
	int nb_lf_thr[3] = {0, 0, 0}, lf_thr[3][4] = {{0}}, nb_qf_thr = 0, qf_thr[4] = {0};
	std::vector<uint8_t> bctx_map;
	int nb_block_ctx = 15;
	static const uint8_t DEFAULT_BLKCTX[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
	if (!custom_bctx) bctx_map.assign(DEFAULT_BLKCTX, DEFAULT_BLKCTX + 39);
	else {
		nb_lf_thr[0] = 1; lf_thr[0][0] = 3;                // X
		nb_lf_thr[1] = 2; lf_thr[1][0] = 60; lf_thr[1][1] = 140;  // Y
		nb_lf_thr[2] = 1; lf_thr[2][0] = 50;               // B
		nb_qf_thr = 2; qf_thr[0] = 4; qf_thr[1] = 9;
		int size = 39 * 2 * 3 * 2 * 3;
		nb_block_ctx = 11;
		bctx_map.resize((size_t) size);
		for (int i = 0; i < size; ++i) bctx_map[(size_t) i] = (uint8_t) ((i * 7 + i / 13) % nb_block_ctx);
		for (int i = 0; i < nb_block_ctx; ++i) bctx_map[(size_t) i] = (uint8_t) i;  // every cluster id present
	}
	const int lfidx_size = (nb_lf_thr[0] + 1) * (nb_lf_thr[1] + 1) * (nb_lf_thr[2] + 1);

	// ---- per LF group: LF image, varblock layout, HF metadata ----
	std::vector<LfGroupW> ggs((size_t) num_lf_groups);
	std::vector<int> forced;  // transform types still to be placed once (coverage)
	if (coverage) for (int t = 0; t < 27; ++t) if (std::max(DCTSEL[t][0], DCTSEL[t][1]) <= max_log) forced.push_back(t);
	// weights of the random mix (8x8-class transforms dominate, like a d1 encode)
	const int mixw[27] = {40, 3, 2, 3, 14, 6, 6, 6, 2, 2, 3, 3, 3, 3, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1};
	for (int ggy = 0, ggi = 0; ggy < ggrows; ++ggy) for (int ggx = 0; ggx < ggcols; ++ggx, ++ggi) {
		LfGroupW &gg = ggs[(size_t) ggi];
		gg.left = ggx * 2048; gg.top = ggy * 2048; gg.w = std::min(2048, W - gg.left); gg.h = std::min(2048, H - gg.top);
		gg.w8 = (gg.w + 7) / 8; gg.h8 = (gg.h + 7) / 8; gg.w64 = (gg.w + 63) / 64; gg.h64 = (gg.h + 63) / 64;
		for (int c = 0; c < 3; ++c) gg.lfq[c] = Channel(gg.w8, gg.h8);
		for (int y = 0; y < gg.h8; ++y) for (int x = 0; x < gg.w8; ++x) {
			float rgb[3]; double xyb[3];
			if (forward) {   // the cell's mean
				for (int c = 0; c < 3; ++c) { double m = 0; for (int j = 0; j < 8; ++j) for (int i = 0; i < 8; ++i) m += plane[c][(size_t) (gg.top + y * 8 + j) * (size_t) PW + (size_t) (gg.left + x * 8 + i)]; xyb[c] = m / 64; }
			} else {
				pic.rgb((float) (gg.left + x * 8) + 3.5f, (float) (gg.top + y * 8) + 3.5f, rgb);
				to_xyb(rgb, xyb);
			}
			gg.lfq[0].at(x, y) = (int32_t) lrint(xyb[1] / lfstep[1]);  // streamed order: Y, X, B
			gg.lfq[1].at(x, y) = (int32_t) lrint(xyb[0] / lfstep[0]);
			// the decoder adds kb_lf * Y to the LF of B and kx_lf * Y to X (j40.h:7115-7116, 7159-7171)
			double yq = (double) gg.lfq[0].at(x, y) * lfstep[1];
			gg.lfq[1].at(x, y) = (int32_t) lrint((xyb[0] - kx_lf * yq) / lfstep[0]);
			gg.lfq[2].at(x, y) = (int32_t) lrint((xyb[2] - kb_lf * yq) / lfstep[2]);
		}
		gg.lfidx.assign((size_t) gg.w8 * (size_t) gg.h8, 0);
		for (int y = 0; y < gg.h8; ++y) for (int x = 0; x < gg.w8; ++x) {  // j40.h:6566-6570
			auto cnt = [&](int v, const int *thr, int n) { int k = 0; for (int t = 0; t < n; ++t) k += v > thr[t]; return k; };
			int xi = cnt(gg.lfq[1].at(x, y), lf_thr[0], nb_lf_thr[0]), yi = cnt(gg.lfq[0].at(x, y), lf_thr[1], nb_lf_thr[1]), bi = cnt(gg.lfq[2].at(x, y), lf_thr[2], nb_lf_thr[2]);
			gg.lfidx[(size_t) y * (size_t) gg.w8 + (size_t) x] = (uint8_t) ((xi * (nb_lf_thr[0] + 1) + bi) * (nb_lf_thr[2] + 1) + yi);
		}
		gg.blocks.assign((size_t) gg.w8 * (size_t) gg.h8, 0);
		for (int y0 = 0; y0 < gg.h8; ++y0) for (int x0 = 0; x0 < gg.w8; ++x0) {
			if (gg.blocks[(size_t) y0 * (size_t) gg.w8 + (size_t) x0]) continue;
			auto fits = [&](int t) {
				int vw8 = 1 << (DCTSEL[t][1] - 3), vh8 = 1 << (DCTSEL[t][0] - 3);
				int x1 = x0 + vw8 - 1, y1 = y0 + vh8 - 1;
				if (x1 >= gg.w8 || y1 >= gg.h8 || (x0 >> 5) != (x1 >> 5) || (y0 >> 5) != (y1 >> 5)) return false;  // j40.h:6659-6660
				for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) if (gg.blocks[(size_t) y * (size_t) gg.w8 + (size_t) x]) return false;
				return true;
			};
			int t = -1;
			for (size_t k = 0; k < forced.size(); ++k) if (fits(forced[k])) { t = forced[k]; forced.erase(forced.begin() + (long) k); break; }
			if (t < 0) {
				int total = 0; for (int k = 0; k < 27; ++k) if (std::max(DCTSEL[k][0], DCTSEL[k][1]) <= max_log) total += mixw[k];
				int pick = (int) rng.below((uint32_t) total);
				for (int k = 0; k < 27; ++k) if (std::max(DCTSEL[k][0], DCTSEL[k][1]) <= max_log) { if (pick < mixw[k]) { t = k; break; } pick -= mixw[k]; }
				if (!fits(t)) t = 0;
			}
			int hfmul_m1 = 3 + (int) rng.below(10);
			if (forward) {
				// what an encoder's heuristics amount to: large transforms only over calm content, the 8x8 specials only where there
				// is something to localise, finer quantisation (larger HfMul) where errors show (calm areas)
				auto worst = [&](int tt) { double a = 0; for (int y = y0; y < y0 + (1 << (DCTSEL[tt][0] - 3)); ++y) for (int x = x0; x < x0 + (1 << (DCTSEL[tt][1] - 3)); ++x) a = std::max(a, cell_activity(gg.left / 8 + x, gg.top / 8 + y)); return a; };
				if (std::max(DCTSEL[t][0], DCTSEL[t][1]) > 3 && worst(t) > 4e-5 * (double) (1 << (12 - DCTSEL[t][0] - DCTSEL[t][1])) ) t = 0;
				if (DCTSEL[t][0] == 3 && DCTSEL[t][1] == 3) {
					const double a = worst(t);
					if (a < 2e-5) t = 0;
					else if (t == 0 && a > 4e-4 && rng.below(3) == 0) { static const int SP[9] = {1, 2, 3, 12, 13, 14, 15, 16, 17}; t = SP[rng.below(9)]; }
				}
				const double a = worst(t);
				hfmul_m1 = (a < 1e-5 ? 8 : a < 1e-4 ? 7 : a < 1e-3 ? 6 : 5) - 1;
			}
			int vw8 = 1 << (DCTSEL[t][1] - 3), vh8 = 1 << (DCTSEL[t][0] - 3), voff = (int) gg.vbs.size();
			for (int y = y0; y < y0 + vh8; ++y) for (int x = x0; x < x0 + vw8; ++x) gg.blocks[(size_t) y * (size_t) gg.w8 + (size_t) x] = 1 << 20 | voff;
			gg.blocks[(size_t) y0 * (size_t) gg.w8 + (size_t) x0] = (t + 2) << 20 | voff;
			gg.vbs.push_back({x0, y0, t, hfmul_m1});
		}
		gg.xfromy = Channel(gg.w64, gg.h64); gg.bfromy = Channel(gg.w64, gg.h64);
		for (auto &v : gg.xfromy.px) v = forward ? 0 : (int32_t) rng.below(9) - 4;
		for (auto &v : gg.bfromy.px) v = forward ? 0 : (int32_t) rng.below(13) - 6;
		gg.blockinfo = Channel((int) gg.vbs.size(), 2);
		for (size_t i = 0; i < gg.vbs.size(); ++i) { gg.blockinfo.at((int) i, 0) = gg.vbs[i].dctsel; gg.blockinfo.at((int) i, 1) = gg.vbs[i].hfmul_m1; }
		gg.sharp = Channel(gg.w8, gg.h8);
		for (auto &v : gg.sharp.px) v = (int32_t) rng.below(8);
	}

	// ---- global MA tree: splits on stream index (property 1) and channel (property 0) ----
	MATree tree;
	if (opt.geti("lftree", 0) == 2) {
		// lftree=2: every LfGroup channel under ONE test of a sample property over two leaves that predict alike (what libjxl's fixed LF
		// trees look like, with other properties and predictors), offsets and multipliers on some
		auto pair = [&](int prop, int thr, int pred, int off = 0, int mshift = 0, int mbits = 0) { int a = tree.leaf(pred, off, mshift, mbits), b = tree.leaf(pred, off, mshift, mbits); return tree.branch(prop, thr, a, b); };
		int lf_y = pair(9, 100, 5), lf_x = pair(8, 0, 4), lf_b = pair(5, 2, 3, 1);
		int lf_xb = tree.branch(0, 1, lf_b, lf_x);
		int lf = tree.branch(0, 0, lf_xb, lf_y);
		int cfl = pair(2, 3, 1), binfo = pair(3, 300, 9, 0, 0, 0), sharp = pair(12, 0, 12, -1);
		int meta_hi = tree.branch(0, 2, sharp, binfo);
		int meta = tree.branch(0, 1, meta_hi, cfl);
		int root = tree.branch(1, 2 * num_lf_groups, meta, lf);
		tree.finalise(root);
	} else if (opt.geti("lftree", 0) == 3) {
		// lftree=3: as 2 with the other testable properties and predictors
		auto pair = [&](int prop, int thr, int pred, int off = 0, int mshift = 0, int mbits = 0) { int a = tree.leaf(pred, off, mshift, mbits), b = tree.leaf(pred, off, mshift, mbits); return tree.branch(prop, thr, a, b); };
		int lf_y = pair(10, 0, 10), lf_x = pair(11, 1, 11), lf_b = pair(14, -1, 8);
		int lf_xb = tree.branch(0, 1, lf_b, lf_x);
		int lf = tree.branch(0, 0, lf_xb, lf_y);
		int cfl = pair(7, 0, 2), binfo = pair(14, 2, 1, 0, 0, 0), sharp = pair(4, 3, 7, 0, 1, 1);
		int meta_hi = tree.branch(0, 2, sharp, binfo);
		int meta = tree.branch(0, 1, meta_hi, cfl);
		int root = tree.branch(1, 2 * num_lf_groups, meta, lf);
		tree.finalise(root);
	} else if (opt.geti("lftree", 0)) {
		// lftree=1: below the stream / channel splits every LfGroup channel gets a subtree of its own over the sample properties, with
		// predictors that look at NE, NEE, NN, NWW -- also the varblock-info channel, whose second row is thousands of samples wide
		int y0 = tree.leaf(13), y1 = tree.leaf(5), y2 = tree.leaf(7), y3 = tree.leaf(12, 1);
		int ya = tree.branch(12, 2, y0, y1), yb = tree.branch(13, 0, y2, y3);   // N - NE, N - NN
		int lf_ysplit = tree.branch(9, 120, ya, yb);                             // W + N - NW
		int lf_x = tree.branch(8, 0, tree.leaf(9), tree.leaf(10));               // W - (WW + NW - NWW)
		int lf_b = tree.branch(14, 1, tree.leaf(11, -1), tree.branch(4, 3, tree.leaf(3), tree.leaf(4)));   // W - WW, |N|
		int lf_xb = tree.branch(0, 1, lf_b, lf_x);
		int lf = tree.branch(0, 0, lf_xb, lf_ysplit);
		int cfl = tree.branch(5, 2, tree.leaf(1), tree.leaf(8));                 // |W|
		int binfo = tree.branch(6, 3, tree.leaf(5), tree.branch(3, 300, tree.leaf(2), tree.branch(2, 0, tree.leaf(13), tree.leaf(0))));   // N, x, y
		int sharp = tree.branch(2, 5, tree.branch(11, 0, tree.leaf(4), tree.leaf(12)), tree.branch(7, 3, tree.leaf(2), tree.leaf(10)));   // y, NW - N, W
		int meta_hi = tree.branch(0, 2, sharp, binfo);
		int meta = tree.branch(0, 1, meta_hi, cfl);
		int root = tree.branch(1, 2 * num_lf_groups, meta, lf);
		tree.finalise(root);
	} else {
		int lf_y = tree.leaf(5), lf_x = tree.leaf(5), lf_b = tree.leaf(4);
		int lf_yhi = tree.leaf(5);
		int lf_ysplit = tree.branch(9, 120, lf_yhi, lf_y);          // W+N-NW > 120
		int lf_xb = tree.branch(0, 1, lf_b, lf_x);
		int lf = tree.branch(0, 0, lf_xb, lf_ysplit);               // channel > 0 ?
		int cfl = tree.leaf(1), binfo = tree.leaf(0), sharp = tree.leaf(2);
		int meta_hi = tree.branch(0, 2, sharp, binfo);
		int meta = tree.branch(0, 1, meta_hi, cfl);
		int root = tree.branch(1, 2 * num_lf_groups, meta, lf);     // sidx > 2*num_lf_groups  <=> HF metadata stream
		tree.finalise(root);
	}
	CodeSpecW treespec; treespec.init(6, std::vector<uint8_t>(6, 0), 1); treespec.log_alpha = 6; treespec.cfg[0] = HybridCfg{4, 1, 0};
	StreamEncoder tree_enc(treespec); tree_tokens(tree, tree_enc); count_stream(treespec, tree_enc);

	CodeSpecW gspec;  // global code spec shared by every LF-group Modular stream
	{
		std::vector<uint8_t> map((size_t) tree.num_ctx);
		for (int i = 0; i < tree.num_ctx; ++i) map[(size_t) i] = (uint8_t) i;
		gspec.init(tree.num_ctx, map, tree.num_ctx);
		gspec.log_alpha = 8;
		for (auto &c : gspec.cfg) c = HybridCfg{4, 2, 0};
	}
	WPParams wpp;
	std::vector<StreamEncoder> lfq_enc, meta_enc;
	for (int ggi = 0; ggi < num_lf_groups; ++ggi) {
		LfGroupW &gg = ggs[(size_t) ggi];
		lfq_enc.emplace_back(gspec); meta_enc.emplace_back(gspec);
		std::vector<Channel> ch{gg.lfq[0], gg.lfq[1], gg.lfq[2]};
		for (int c = 0; c < 3; ++c) encode_channel(tree, ch, c, 1 + ggi, wpp, lfq_enc.back());
		std::vector<Channel> mc{gg.xfromy, gg.bfromy, gg.blockinfo, gg.sharp};
		for (int c = 0; c < 4; ++c) encode_channel(tree, mc, c, 1 + 2 * num_lf_groups + ggi, wpp, meta_enc.back());
		count_stream(gspec, lfq_enc.back()); count_stream(gspec, meta_enc.back());
	}

	// dq=2: two of the larger matrices in raw form (DCT32x32 and DCT8x16), coded with the global tree and code spec
	std::vector<std::pair<int, StreamEncoder>> dq_raw;
	if (opt.geti("dq", 0) >= 2) for (int idx : {5, 6}) {
		const int ncols = idx == 5 ? 32 : 16, nrows = idx == 5 ? 32 : 8;
		std::vector<Channel> mc;
		for (int c = 0; c < 3; ++c) { Channel m(ncols, nrows); for (int y = 0; y < nrows; ++y) for (int x = 0; x < ncols; ++x) m.at(x, y) = 300 + 90 * (x + y) * (c + 1) + 17 * ((x * 5 + y * 3 + c) & 7); mc.push_back(m); }
		dq_raw.emplace_back(idx, StreamEncoder(gspec));
		for (int c = 0; c < 3; ++c) encode_channel(tree, mc, c, 1 + 3 * num_lf_groups + idx, wpp, dq_raw.back().second);
		count_stream(gspec, dq_raw.back().second);
	}

	// ---- optional alpha extra channel: a Modular sub-image per pass group, coded after the group's HF coefficients
	//      (j40.h:7024-7034) with the global tree and code spec ----
	const int with_alpha = opt.geti("alpha", 0);
	if (with_alpha && num_passes > 1) die("vardct: alpha with several passes is not generated");
	std::vector<StreamEncoder> alpha_enc;
	if (with_alpha) {
		Channel alpha(W, H);
		for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
			int v = (x * 3 + y * 2) / 4 + (int) rng.below(5) - 2 + (((x / 40) + (y / 24)) % 3 == 0 ? 90 : 0);
			alpha.at(x, y) = std::max(0, std::min(255, v));
		}
		for (int g = 0; g < num_groups; ++g) {
			const int gx = (g % gcols) * 256, gy = (g / gcols) * 256, gw = std::min(256, W - gx), gh = std::min(256, H - gy);
			std::vector<Channel> sub(1, Channel(gw, gh));
			for (int y = 0; y < gh; ++y) for (int x = 0; x < gw; ++x) sub[0].at(x, y) = alpha.at(gx + x, gy + y);
			alpha_enc.emplace_back(gspec);
			encode_channel(tree, sub, 0, 1 + 3 * num_lf_groups + 17 + g, wpp, alpha_enc.back());
			count_stream(gspec, alpha_enc.back());
		}
	}

	// ---- HF coefficients: tokens per (pass, group) with the decoder's context model ----
	const int ctx_per_preset = 495 * nb_block_ctx;
	const int hf_prefix = opt.geti("hfprefix", 0);          // HF coefficient streams with prefix codes instead of rANS
	const int hf_lz77 = opt.geti("hflz77", 0);              // ... with LZ77 copies for runs of equal small values
	std::vector<CodeSpecW> cspec((size_t) num_passes);
	std::vector<std::vector<StreamEncoder>> hf_enc((size_t) num_passes);
	std::vector<int> group_preset((size_t) num_groups);
	for (int g = 0; g < num_groups; ++g) group_preset[(size_t) g] = num_presets > 1 ? (int) rng.below((uint32_t) num_presets) : 0;
	for (int pass = 0; pass < num_passes; ++pass) {
		CodeSpecW &cs = cspec[(size_t) pass];
		const int nctx = ctx_per_preset * num_presets;
		std::vector<uint8_t> map((size_t) nctx);
		int nclusters = 0;
		for (int ctx = 0; ctx < nctx; ++ctx) {
			int preset = ctx / ctx_per_preset, r = ctx % ctx_per_preset, cl;
			if (r < 37 * nb_block_ctx) {            // number-of-nonzeros contexts
				int k = r / nb_block_ctx, b = r % nb_block_ctx;
				cl = small_clusters ? (k < 8 ? 0 : 1) : (k < 4 ? 0 : k < 12 ? 1 : k < 24 ? 2 : 3) * 3 + b % 3;
			} else {                                // coefficient contexts
				int q = r - 37 * nb_block_ctx, b = q / 458, s = q % 458;
				cl = small_clusters ? 2 + std::min(5, s / 80) : 12 + (s / 46) * 3 + b % 3;
			}
			if (!small_clusters && num_presets > 1 && preset == 1) cl = (cl * 5 + 3) % 42;  // presets share clusters differently
			map[(size_t) ctx] = (uint8_t) cl; nclusters = std::max(nclusters, cl + 1);
		}
		{   // cluster ids must be dense (j40.h:2584-2588): compact them
			std::vector<int> remap((size_t) nclusters, -1); int next = 0;
			for (auto &m : map) { if (remap[m] < 0) remap[m] = next++; m = (uint8_t) remap[m]; }
			nclusters = next;
		}
		if (hf_lz77) { map.push_back((uint8_t) nclusters); ++nclusters; cs.lz77 = true; }   // the distance context gets its own cluster
		cs.init(nctx, map, nclusters);
		cs.lz_min_symbol = 224; cs.lz_min_length = 3; cs.lz_len_cfg = HybridCfg{0, 0, 0};
		cs.use_prefix = hf_prefix != 0;
		cs.log_alpha = (hf_lz77 || hf_prefix) ? 8 : log_alpha;
		for (int c = 0; c < nclusters; ++c) cs.cfg[(size_t) c] = (c & 1) && !hf_lz77 ? HybridCfg{4, 1, 1} : HybridCfg{4, 2, 0};
		hf_enc[(size_t) pass].reserve((size_t) num_groups);
	}
	// custom coefficient orders: Lehmer codes per (pass, order, channel); the writer only needs them
	// in the header because coefficients are synthesised in scan-index space
	const int LOG_ORDER_SIZE[13][2] = {{3,3},{3,3},{4,4},{5,5},{3,4},{3,5},{4,5},{6,6},{5,6},{7,7},{6,7},{8,8},{7,8}};
	std::vector<std::vector<std::vector<uint32_t>>> lehmer((size_t) num_passes);  // [pass][order*3+c] -> values
	int used_orders_mask = 0;
	if (custom_orders) {
		used_orders_mask = (1 << 0) | (1 << 2) | (1 << 4);
		for (int pass = 0; pass < num_passes; ++pass) {
			lehmer[(size_t) pass].assign(13 * 3, {});
			for (int o = 0; o < 13; ++o) if (used_orders_mask >> o & 1) for (int c = 0; c < 3; ++c) {
				int size = 1 << (LOG_ORDER_SIZE[o][0] + LOG_ORDER_SIZE[o][1]), skip = size / 64;
				int end = 6 + (int) rng.below(10);
				std::vector<uint32_t> v;
				for (int i = 0; i < end; ++i) v.push_back(rng.below((uint32_t) std::min(24, size - skip - i)));
				lehmer[(size_t) pass][(size_t) (o * 3 + c)] = v;
			}
		}
	}

	std::vector<float> fpix, fcoef, fydeq;   // forward=1: one block's samples, coefficients, dequantised Y coefficients
	for (int pass = 0; pass < num_passes; ++pass) {
		for (int g = 0; g < num_groups; ++g) {
			hf_enc[(size_t) pass].emplace_back(cspec[(size_t) pass]);
			StreamEncoder &enc = hf_enc[(size_t) pass].back();
			const int grow = g / gcols, gcol = g % gcols, ggi = (grow / 8) * ggcols + gcol / 8;
			const LfGroupW &gg = ggs[(size_t) ggi];
			const int gx8 = (gcol % 8) * 32, gy8 = (grow % 8) * 32;
			const int gw = std::min(W, (gcol + 1) * 256) - gcol * 256, gh = std::min(H, (grow + 1) * 256) - grow * 256;
			const int gw8 = (gw + 7) / 8, gh8 = (gh + 7) / 8;
			const int ctxoff = ctx_per_preset * group_preset[(size_t) g];
			std::vector<std::array<int8_t, 3>> nonzeros((size_t) gw8 * (size_t) gh8, std::array<int8_t, 3>{0, 0, 0});
			for (int y8 = 0; y8 < gh8; ++y8) for (int x8 = 0; x8 < gw8; ++x8) {
				int cell = gg.blocks[(size_t) (gy8 + y8) * (size_t) gg.w8 + (size_t) (gx8 + x8)];
				int dctsel = cell >> 20;
				if (dctsel < 2) continue;
				dctsel -= 2;
				const LfGroupW::VB &vb = gg.vbs[(size_t) (cell & 0xfffff)];
				const int log_rows = DCTSEL[dctsel][0], log_cols = DCTSEL[dctsel][1], log_size = log_rows + log_cols, order_idx = DCTSEL[dctsel][2];
				int qfidx = 0; for (int j = 0; j < nb_qf_thr; ++j) qfidx += vb.hfmul_m1 >= qf_thr[j];
				int lfidx = gg.lfidx[(size_t) (gy8 + y8) * (size_t) gg.w8 + (size_t) (gx8 + x8)];
				int bctx0 = (order_idx * (nb_qf_thr + 1) + qfidx) * lfidx_size + lfidx, bctxc = 13 * (nb_qf_thr + 1) * lfidx_size;
				const int nzpos = y8 * gw8 + x8;
				for (int cyxb = 0; cyxb < 3; ++cyxb) {
					const int c = cyxb == 0 ? 1 : cyxb == 1 ? 0 : 2;
					const int bctx = bctx_map[(size_t) (bctx0 + bctxc * cyxb)];
					// choose the non-zero positions: Bernoulli with frequency decay; X/B sparser than Y
					const int size = 1 << log_size, first = size >> 6;
					double p = density * (c == 1 ? 1.0 : c == 0 ? 0.35 : 0.55) / (double) num_passes;
					std::vector<std::pair<int, int>> coefs;  // (scan index, value)
					double pk = p; const double dk = pow(decay, 64.0 / (double) size);
					if (forward) {
						// the block's samples -> coefficients -> quantised with the weights the decoder divides by (j40.h:7086-7094);
						// X and B code what is left after the decoder's chroma-from-luma term (default factors: 0 and 1 times the
						// dequantised Y coefficient, j40.h:7138-7143, 7159-7171)
						const int R = 1 << log_rows, C = 1 << log_cols;
						const int px0 = gg.left + (gx8 + x8) * 8, py0 = gg.top + (gy8 + y8) * 8;
						fpix.resize((size_t) size); fcoef.resize((size_t) size);
						for (int y = 0; y < R; ++y) for (int x = 0; x < C; ++x) fpix[(size_t) (y * C + x)] = plane[c][(size_t) (py0 + y) * (size_t) PW + (size_t) (px0 + x)];
						fwdx->analyse(dctsel, fpix.data(), fcoef.data());
						const float mult1 = 65536.0f / (float) global_scale / (float) (vb.hfmul_m1 + 1);
						const float mult_c = c == 1 ? mult1 : c == 0 ? mult1 * powf(0.8f, (float) (x_qm - 2)) : mult1 * powf(0.8f, (float) (b_qm - 2));
						const std::vector<float> &wt = fwdx->weight[dctsel][c];
						const std::vector<int32_t> &ord = fwdx->order[dctsel];
						if (c == 1) fydeq.assign((size_t) size, 0.0f);
						const float dead = (float) opt.getd("deadzone", 0.56);
						for (int i = first; i < size; ++i) {
							const int pos = ord[(size_t) i];
							float target = fcoef[(size_t) pos];
							if (c == 2) target -= fydeq[(size_t) pos];
							const float scaled = target * wt[(size_t) pos] / mult_c;
							const int q = fabsf(scaled) < dead ? 0 : (int) lrintf(scaled);
							if (c == 1 && q) fydeq[(size_t) pos] = synthfwd::dequant((float) q, 1, mult_c, wt[(size_t) pos]);
							if (q) coefs.push_back({i, q});
						}
					} else {
					// d1-like statistics: non-zeros concentrate at low frequencies (so the scan ends early, as an
					// encoder's would) and low frequencies carry the larger magnitudes
					double cont = opt.getd("cont", 0.86);
					for (int i = first; i < size && pk > 2e-3; ++i) {
						if (rng.unit() < pk) {
							int mag = 1; while (mag < 40 && rng.unit() < cont) ++mag;
							coefs.push_back({i, rng.below(2) ? mag : -mag});
						}
						pk *= dk; cont = 0.3 + (cont - 0.3) * pow(dk, 0.6);
					}
					}
					int nz = (int) coefs.size();
					if (nz > (63 << (log_size - 6))) { coefs.resize((size_t) (63 << (log_size - 6))); nz = (int) coefs.size(); }
					int pred;
					if (x8 > 0) pred = y8 > 0 ? (nonzeros[(size_t) nzpos - 1][(size_t) c] + nonzeros[(size_t) (nzpos - gw8)][(size_t) c] + 1) >> 1 : nonzeros[(size_t) nzpos - 1][(size_t) c];
					else pred = y8 > 0 ? nonzeros[(size_t) (nzpos - gw8)][(size_t) c] : 32;
					int nzctx = ctxoff + bctx + (pred < 8 ? pred : 4 + pred / 2) * nb_block_ctx;
					enc.add((uint32_t) nzctx, (uint32_t) nz);
					int qnz = (nz + (1 << (log_size - 6)) - 1) >> (log_size - 6);
					for (int i = 0; i < (1 << (log_rows - 3)); ++i) for (int j = 0; j < (1 << (log_cols - 3)); ++j)
						if (y8 + i < gh8 && x8 + j < gw8) nonzeros[(size_t) (nzpos + i * gw8 + j)][(size_t) c] = (int8_t) qnz;
					const int cctx = ctxoff + 458 * bctx + 37 * nb_block_ctx;
					int prev = nz <= (1 << (log_size - 4)), remaining = nz;
					size_t next = 0;
					for (int i = first; remaining > 0 && i < size; ++i) {
						int ctx = cctx + nnz_ctx2((remaining + (1 << (log_size - 6)) - 1) >> (log_size - 6)) + FREQ_CTX2[i >> (log_size - 6)] + prev;
						int v = 0;
						if (next < coefs.size() && coefs[next].first == i) v = coefs[next++].second;
						enc.add((uint32_t) ctx, pack_signed(v));
						prev = v != 0; remaining -= prev;
					}
				}
			}
			if (hf_lz77) {
				// run-length pass: a run of >= 3 identical small values after its first occurrence becomes one copy with
				// distance 1 (dist_mult is 0 for coefficient streams, so the coded distance value is distance - 1, j40.h:2851)
				const CodeSpecW &cs = cspec[(size_t) pass];
				std::vector<StreamEncoder::Item> out;
				const auto &it = enc.items;
				for (size_t i = 0; i < it.size(); ) {
					size_t j = i + 1;
					while (j < it.size() && it[j].token == it[i].token && it[i].nextra == 0 && it[j].nextra == 0 && it[i].token < 16) ++j;
					out.push_back(it[i]);
					const size_t run = j - i - 1;
					if (run >= 3 && it[i].token < 16) {
						HToken t = hybrid_encode((uint32_t) run - (uint32_t) cs.lz_min_length, cs.lz_len_cfg);
						out.push_back({it[i + 1].cluster, t.token + (uint32_t) cs.lz_min_symbol, t.extra, (uint8_t) t.nextra});
						const uint32_t lzcl = cs.cluster_map[(size_t) cs.total_dist() - 1];
						HToken d = hybrid_encode(0, cs.cfg[lzcl]);
						out.push_back({lzcl, d.token, d.extra, (uint8_t) d.nextra});
						i = j;
					} else i = i + 1;
				}
				enc.items.swap(out);
			}
			count_stream(cspec[(size_t) pass], enc);
		}
	}

	// ---- assemble sections ----
	std::vector<std::vector<uint8_t>> sections;
	{   // LfGlobal (j40.h:6257)
		BitWriter bw;
		bw.put(1, 1);                                            // LF channel dequantisation: default
		bw.u32(global_scale, 1, 11, 2049, 11, 4097, 12, 8193, 16);
		bw.u32(quant_lf, 16, 0, 1, 5, 1, 8, 1, 16);
		if (!custom_bctx) bw.put(1, 1);
		else {
			bw.put(0, 1);
			for (int i = 0; i < 3; ++i) {
				bw.put((uint64_t) nb_lf_thr[i], 4);
				for (int j = 0; j < nb_lf_thr[i]; ++j) bw.u32(pack_signed(lf_thr[i][j]), 0, 4, 16, 8, 272, 16, 65808, 32);
			}
			bw.put((uint64_t) nb_qf_thr, 4);
			for (int j = 0; j < nb_qf_thr; ++j) bw.u32(qf_thr[j] - 1, 0, 2, 4, 3, 12, 5, 44, 8);
			write_cluster_map(bw, bctx_map, nb_block_ctx);
		}
		if (!custom_cfl) bw.put(1, 1);                   // LF channel correlation: default
		else { bw.put(0, 1); bw.u32(128, 84, 0, 256, 0, 2, 8, 258, 16); bw.f16(0.125f); bw.f16(0.75f); bw.put(127 + 3, 8); bw.put(127 - 2, 8); }
		bw.put(1, 1);                                            // global tree present
		write_code_spec(bw, treespec); tree_enc.flush(bw);
		write_code_spec(bw, gspec);
		if (with_alpha) {   // the global Modular image (extra channels only): header, no channel decoded here (j40.h:6329-6338)
			write_modular_header(bw, true, nullptr, {});
			StreamEncoder none(gspec); none.flush(bw);
		}
		bw.pad();
		sections.push_back(bw.bytes);
	}
	for (int ggi = 0; ggi < num_lf_groups; ++ggi) {  // LfGroup (j40.h:6722)
		LfGroupW &gg = ggs[(size_t) ggi];
		BitWriter bw;
		bw.put(0, 2);                                            // extra_precision
		write_modular_header(bw, true, nullptr, {});
		lfq_enc[(size_t) ggi].flush(bw);
		bw.put((uint64_t) (gg.vbs.size() - 1), ceil_lg((uint32_t) (gg.w8 * gg.h8)));
		write_modular_header(bw, true, nullptr, {});
		meta_enc[(size_t) ggi].flush(bw);
		bw.pad();
		sections.push_back(bw.bytes);
	}
	{   // HfGlobal + HfPass (j40.h:6819)
		BitWriter bw;
		if (opt.geti("dq", 0)) { bw.put(0, 1); write_dq_matrices(bw, dq_raw); }
		else bw.put(1, 1);                                       // all dequantisation matrices default
		bw.put((uint64_t) (num_presets - 1), ceil_lg((uint32_t) num_groups));
		for (int pass = 0; pass < num_passes; ++pass) {
			if (!used_orders_mask) bw.put(2, 2);                 // used_orders = 0
			else {
				bw.put(3, 2); bw.put((uint64_t) used_orders_mask, 13);
				CodeSpecW ospec; ospec.init(8, std::vector<uint8_t>(8, 0), 1); ospec.log_alpha = 6; ospec.cfg[0] = HybridCfg{4, 1, 0};
				StreamEncoder oenc(ospec);
				for (int o = 0; o < 13; ++o) if (used_orders_mask >> o & 1) for (int c = 0; c < 3; ++c) {
					const auto &v = lehmer[(size_t) pass][(size_t) (o * 3 + c)];
					int size = 1 << (LOG_ORDER_SIZE[o][0] + LOG_ORDER_SIZE[o][1]);
					oenc.add((uint32_t) std::min(7, ceil_lg((uint32_t) size + 1)), (uint32_t) v.size());  // j40.h:5437
					uint32_t prev = 0;
					for (uint32_t x : v) { oenc.add((uint32_t) std::min(7, ceil_lg(prev + 1)), x); prev = x; }
				}
				count_stream(ospec, oenc);
				write_code_spec(bw, ospec); oenc.flush(bw);
			}
			write_code_spec(bw, cspec[(size_t) pass]);
		}
		bw.pad();
		sections.push_back(bw.bytes);
	}
	for (int pass = 0; pass < num_passes; ++pass) for (int g = 0; g < num_groups; ++g) {  // PassGroup (j40.h:7007)
		BitWriter bw;
		bw.put((uint64_t) group_preset[(size_t) g], ceil_lg((uint32_t) num_presets));
		hf_enc[(size_t) pass][(size_t) g].flush(bw);
		if (with_alpha) { write_modular_header(bw, true, nullptr, {}); alpha_enc[(size_t) g].flush(bw); }
		bw.pad();
		sections.push_back(bw.bytes);
	}

	// ---- codestream ----
	BitWriter cs;
	cs.put(0xff, 8); cs.put(0x0a, 8);
	write_size_header(cs, W, H);
	const int icc_bytes = opt.geti("icc", 0);                // > 0: ColourEncoding with want_icc and an ICC stream of that many coded bytes
	const int img_bpp = opt.geti("bpp", 8);   // 9..15: the renderer's scaling to 8 bits and the long way through the transfer curve get work
	if (img_bpp != 8 && (icc_bytes || with_alpha)) die("vardct: bpp combines with neither icc nor alpha here");
	const int noxyb = opt.geti("noxyb", 0);
	if (noxyb && (!with_alpha || icc_bytes || !nonzero_header || x_qm != 3 || b_qm != 2)) die("vardct: noxyb wants alpha=1 fullheader=1 and the default qm scales");
	if (img_bpp != 8) {
		cs.put(0, 1);                       // ImageMetadata: not all_default
		cs.put(0, 1);                       // no extra fields
		write_bit_depth(cs, img_bpp);
		cs.put(1, 1);                       // modular_16bit_buffers
		cs.put(0, 2);                       // no extra channels
		cs.put(1, 1);                       // xyb_encoded
		cs.put(1, 1);                       // ColourEncoding.all_default
		cs.put(0, 2);                       // extensions
		cs.put(1, 1);                       // default_m
	} else if (!icc_bytes && !with_alpha) {
		cs.put(1, 1);   // ImageMetadata.all_default: 8-bit, XYB, no extra channels
		cs.put(1, 1);   // default_m
	} else if (!icc_bytes) {
		cs.put(0, 1);                       // ImageMetadata: not all_default
		cs.put(0, 1);                       // no extra fields
		cs.put(0, 1); cs.put(0, 2);         // integer samples, 8 bits
		cs.put(1, 1);                       // modular_16bit_buffers
		cs.put(1, 2); cs.put(1, 1);         // one extra channel, d_alpha
		cs.put(noxyb ? 0 : 1, 1);           // xyb_encoded (noxyb=1: the reference runs the XYB inverse on VarDCT frames regardless, j40.h:7206)
		cs.put(1, 1);                       // ColourEncoding.all_default
		cs.put(0, 2);                       // extensions
		cs.put(1, 1);                       // default_m
	} else {
		cs.put(0, 1);                       // ImageMetadata: not all_default
		cs.put(0, 1);                       // no extra fields
		cs.put(0, 1); cs.put(0, 2);         // integer samples, 8 bits
		cs.put(1, 1);                       // modular_16bit_buffers
		cs.put(0, 2);                       // no extra channels
		cs.put(1, 1);                       // xyb_encoded
		cs.put(0, 1);                       // ColourEncoding: not all_default
		cs.put(1, 1);                       // want_icc
		cs.put(0, 2);                       // colour_space = RGB (enum selector 0)
		cs.put(0, 2);                       // extensions
		cs.put(1, 1);                       // default_m
		write_icc_stream(cs, rng, icc_bytes);
	}
	cs.pad();       // frame header starts byte aligned (j40.h:5228)
	if (!nonzero_header) cs.put(1, 1);  // FrameHeader.all_default
	else {
		cs.put(0, 1);
		cs.put(0, 2);                       // regular frame
		cs.put(0, 1);                       // VarDCT
		cs.u64(skip_smooth ? 128 : 0);      // flags
		if (noxyb) cs.put(0, 1);            // do_ycbcr (read only without xyb_encoded)
		cs.put(0, 2);                       // log_upsampling
		for (int i = 0; i < with_alpha; ++i) cs.put(0, 2);   // ec_log_upsampling
		if (!noxyb) { cs.put((uint64_t) x_qm, 3); cs.put((uint64_t) b_qm, 3); }
		cs.u32(num_passes, 1, 0, 2, 0, 3, 0, 4, 3);
		if (num_passes > 1) {
			cs.u32(0, 0, 0, 1, 0, 2, 0, 3, 1);                  // num_ds = 0
			for (int i = 0; i < num_passes - 1; ++i) cs.put(0, 2);  // shift[i]
		}
		cs.put(0, 1);                       // have_crop
		for (int i = -1; i < with_alpha; ++i) cs.u32(0, 0, 0, 1, 0, 2, 0, 3, 2);  // blend mode: replace (the frame's, then each extra channel's, j40.h:5299)
		cs.put(1, 1);                       // is_last
		cs.u32(0, 0, 0, 0, 4, 16, 5, 48, 10);  // name length 0
		// RestorationFilter (j40.h:5339-5366). gab=1: Gaborish with the default weights, gab=2: custom weights; epf=1..3: iterations of the
		// edge-preserving filter, always with a custom sharpness table (the default table's first entry is 0, which the reference's
		// j40__epf_recip_sigmas rejects with "epf0", j40.h:7384; epflut=0 writes the default all the same), epfw=1: custom channel
		// scales, epfs=1: custom sigma parameters; rfdefault=1: all_default = 1 (the reference then still reads the conditional bits)
		const int gab = opt.geti("gab", 0), epf = opt.geti("epf", 0);
		if (opt.geti("rfdefault", 0)) { cs.put(1, 1); cs.put(0, 1); cs.put(0, 1); cs.put(0, 1); cs.put(0, 1); }   // all_default; then gab_custom 0, sharp_custom 0, weight_custom 0, sigma_custom 0
		else {
			cs.put(0, 1);                       // restoration: !all_default
			cs.put(gab ? 1 : 0, 1);
			if (gab) {
				cs.put(gab == 2 ? 1 : 0, 1);
				if (gab == 2) { const float w[6] = {0.125f, 0.0625f, 0.1875f, 0.03125f, 0.09375f, 0.078125f}; for (float v : w) cs.f16(v); }
			}
			cs.put((uint64_t) epf, 2);
			if (epf) {
				const int lut = opt.geti("epflut", 1);
				cs.put(lut ? 1 : 0, 1);
				if (lut) { const float t[8] = {0.25f, 0.375f, 0.5f, 0.625f, 0.75f, 1.0f, 1.25f, 1.5f}; for (float v : t) cs.f16(v); }
				const int ew = opt.geti("epfw", 0);
				cs.put(ew ? 1 : 0, 1);
				if (ew) { cs.f16(32.0f); cs.f16(6.0f); cs.f16(3.0f); cs.put(0, 32); }
				const int es = opt.geti("epfs", 0);
				cs.put(es ? 1 : 0, 1);
				if (es) { cs.f16(0.5f); cs.f16(0.75f); cs.f16(5.0f); cs.f16(0.625f); }
			}
			cs.u64(0);                          // restoration extensions
		}
		cs.u64(0);                          // frame extensions
	}
	// TOC (j40.h:5505-5531)
	write_toc_and_sections(cs, sections, opt.geti("permute", 0), rng, opt.geti("slack", 0));

	std::vector<uint8_t> file;
	if (!container) file = cs.bytes;
	else {
		// ISOBMFF wrapping (j40.h:1479): signature box, ftyp, then the codestream split over two jxlp boxes
		static const uint8_t HEAD[32] = {0, 0, 0, 0x0c, 'J', 'X', 'L', ' ', 0x0d, 0x0a, 0x87, 0x0a, 0, 0, 0, 0x14, 'f', 't', 'y', 'p', 'j', 'x', 'l', ' ', 0, 0, 0, 0, 'j', 'x', 'l', ' '};
		file.assign(HEAD, HEAD + 32);
		auto box = [&](const char *type, const uint8_t *p, size_t n, int jxlp_index) {
			size_t total = 8 + n + (jxlp_index != -1 ? 4 : 0);
			uint8_t hd[8] = {(uint8_t) (total >> 24), (uint8_t) (total >> 16), (uint8_t) (total >> 8), (uint8_t) total, (uint8_t) type[0], (uint8_t) type[1], (uint8_t) type[2], (uint8_t) type[3]};
			file.insert(file.end(), hd, hd + 8);
			if (jxlp_index != -1) { uint32_t idx = (uint32_t) jxlp_index; uint8_t ix[4] = {(uint8_t) (idx >> 24), (uint8_t) (idx >> 16), (uint8_t) (idx >> 8), (uint8_t) idx}; file.insert(file.end(), ix, ix + 4); }
			file.insert(file.end(), p, p + n);
		};
		if (container == 1) box("jxlc", cs.bytes.data(), cs.bytes.size(), -1);
		else {
			size_t half = cs.bytes.size() / 3;
			// NOTE the reference treats a jxlp index *without* the top bit as "last" (j40.h:1557, inverted
			// w.r.t. ISO 18181-2); order the flags the way it accepts
			box("jxlp", cs.bytes.data(), half, (int) 0x80000000u);
			static const uint8_t junk[5] = {1, 2, 3, 4, 5};
			box("xml ", junk, 5, -1);
			box("jxlp", cs.bytes.data() + half, cs.bytes.size() - half, 1);
		}
	}
	if (!write_file(out, file)) die("cannot write output");
	double bpp = 8.0 * (double) file.size() / ((double) W * H);
	fprintf(stderr, "vardct %dx%d: %zu bytes (%.3f bpp), %d groups, %d LF groups, %d passes\n", W, H, file.size(), bpp, num_groups, num_lf_groups, num_passes);
	return 0;
}

// ------------------------------------------------------------------------------------------------
// Modular frames (lossless-style): RGB or RGBA, global RCT and/or Palette, MA tree with a choice of
// predictors (incl. the weighted predictor), ANS or prefix codes, optional LZ77 run-lengths.
// Single-group frames put everything into LfGlobal (one section, j40.h:6329-6336); larger frames
// code every group's rectangle in its own PassGroup section (j40.h:7024-7033).

static void forward_rct(std::vector<Channel> &ch, int first, int type) {
	// exact inverse of the decoder's RCT for the un-permuted types (j40.h:4341-4393)
	const size_t n = ch[(size_t) first].px.size();
	int32_t *a = ch[(size_t) first].px.data(), *b = ch[(size_t) first + 1].px.data(), *c = ch[(size_t) first + 2].px.data();
	for (size_t i = 0; i < n; ++i) {
		int32_t p0 = a[i], p1 = b[i], p2 = c[i];
		switch (type % 7) {
		case 0: break;
		case 1: c[i] = p2 - p0; break;
		case 2: c[i] = p2 - p0; b[i] = p1; break;   // decoder: out2 = in1 + in0 (needs in1 == in2 - in0; see below)
		case 3: b[i] = p1 - p0; c[i] = p2 - p0; break;
		case 4: b[i] = p1 - ((p0 >> 1) + (p2 >> 1) + (p0 & p2 & 1)); break;
		case 5: c[i] = p2 - p0; b[i] = p1 - p0 - ((p2 - p0) >> 1); break;
		case 6: { int32_t co = p0 - p2, tmp = p2 + (co >> 1), cg = p1 - tmp, y = tmp + (cg >> 1); a[i] = y; b[i] = co; c[i] = cg; break; }
		}
	}
}

int run_modular(int W, int H, uint64_t seed, const char *out, const Options &opt) {
	SplitMix64 rng(seed * 0x9e3779b97f4a7c15ull + 777);
	const int alpha = opt.geti("alpha", 0);
	const int group_shift = opt.geti("groupshift", 8);
	const int rct = opt.geti("rct", 6);                   // -1: none
	const int use_prefix = opt.geti("prefix", 0);
	const int lz77 = opt.geti("lz77", 0);
	const int tree_kind = opt.geti("tree", 0);            // 0 gradient, 1 per-channel leaves + property splits, 2 weighted predictor, 3 previous-channel properties
	const int palette = opt.geti("palette", 0);           // 0 none, 1 plain palette, 2 with deltas / synthetic colours, 3 with delta prediction
	const int container = opt.geti("container", 0);
	const int bpp = opt.geti("bpp", 8);   // 8..15 (16-bit buffers)
	if (bpp < 8 || bpp > 15) die("modular: bpp 8..15");
	if (bpp != 8 && alpha) die("modular: the default alpha channel has 8 bits; the reference refuses a different colour depth");
	const int gdim = 1 << group_shift;
	// repeat=K: the frame is the (W/K) x (H/K) picture tiled K x K times. Only the base picture is synthesised and encoded; its
	// group sections are reused (no tree here looks at the stream index), which makes 16384 x 16384 streams cheap to write
	const int num_passes = opt.geti("passes", 1);
	const int repeat = opt.geti("repeat", 1);
	if (repeat > 1 && num_passes > 1) die("repeat and passes do not combine");
	// squeeze=1: a Squeeze transform with the default parameter list behind the RCT (the "progressive" lossless form); 2: the same list
	// written out explicitly; 3: a short explicit list with residual channels appended (not in place) and partial channel ranges.
	// Squeeze couples neighbouring groups, so with repeat=K the base picture is tiled first and the whole frame is transformed;
	// group sections with identical content (most of them, the picture being periodic) are encoded once and shared.
	const int squeeze = opt.geti("squeeze", 0);
	if (squeeze && (num_passes > 1 || opt.geti("palette", 0) || opt.geti("localrct", -1) >= 0 || opt.geti("localpalette", 0) || opt.geti("localtree", 0))) die("squeeze combines with rct / tree / prefix / lz77 / alpha / extra / bpp / repeat only");
	const int Wfull = W, Hfull = H;
	const int tile_w = W / repeat, tile_h = H / repeat;   // the picture that is synthesised
	if (repeat > 1 && squeeze) {
		if (W % repeat || H % repeat) die("repeat: the frame must be whole tiles");
	} else if (repeat > 1) {
		if (W % (repeat * gdim) || H % (repeat * gdim)) die("repeat: the base picture must be whole groups");
		W /= repeat; H /= repeat;
		if (W == gdim && H == gdim) die("repeat: the base picture must have more than one group");
	}
	const int gcols = (W + gdim - 1) / gdim, grows = (H + gdim - 1) / gdim, num_groups = gcols * grows;
	const int num_lf_groups = ((W + 8 * gdim - 1) / (8 * gdim)) * ((H + 8 * gdim - 1) / (8 * gdim));
	const bool single = num_groups == 1;
	if (single && opt.geti("passes", 1) > 1) die("modular: one group with several passes is not generated (the frame would not be a single section)");
	// extra=K: K more extra channels (type depth, same bit depth) ahead of the alpha channel, so that alpha is not the first one
	const int extra = opt.geti("extra", 0);
	if (extra < 0 || extra + (alpha ? 1 : 0) > 4) die("modular: at most four extra channels (j40.h:3247)");
	const int ncolour = 3, nch = ncolour + extra + (alpha ? 1 : 0);

	// ---- source picture -> channels: colour first, then extra channels (the renderer takes channels
	//      0..2 as RGB and 3.. as extra channels, j40.h:7923-7936) ----
	const int pic_w = squeeze ? tile_w : W, pic_h = squeeze ? tile_h : H;
	Picture pic(pic_w, pic_h, seed);
	std::vector<Channel> ch;
	for (int c = 0; c < nch; ++c) ch.emplace_back(pic_w, pic_h);
	const int colour0 = 0;               // index of the first colour channel
	for (int y = 0; y < pic_h; ++y) for (int x = 0; x < pic_w; ++x) {
		float rgb[3]; pic.rgb((float) x, (float) y, rgb);
		// flat regions + a bit of texture so that run-lengths and the predictors both get work
		for (int c = 0; c < 3; ++c) {
			int v = (int) (rgb[c] * 255.0f);
			v = (v / 6) * 6 + (int) (Picture::hash01((uint64_t) x * 7919 + (uint64_t) y * 104729 + (uint64_t) c + seed) < 0.08f);
			ch[(size_t) (colour0 + c)].at(x, y) = std::min(255, std::max(0, v)) * ((1 << bpp) - 1) / 255;
		}
		for (int k = 0; k < extra; ++k) ch[(size_t) (3 + k)].at(x, y) = (((x >> 3) * (k + 2) + (y >> 2)) & 31) * ((1 << bpp) - 1) / 31;
		if (alpha) ch[(size_t) (3 + extra)].at(x, y) = ((x / 37 + y / 29) & 3) == 0 ? 128 + ((x * 3 + y) & 63) : 255;
	}

	if (squeeze && repeat > 1) {   // lay the tile out over the whole frame
		for (Channel &c : ch) {
			Channel full(W, H);
			for (int y = 0; y < H; ++y) for (int x0 = 0; x0 < W; x0 += tile_w) memcpy(&full.at(x0, y), &c.at(0, y % tile_h), sizeof(int32_t) * (size_t) tile_w);
			c = std::move(full);
		}
	}

	// fourvalues=1: samples take four values only; with the zero predictor (tree=4) and no RCT every residual token is one of four
	// symbols, which makes prefix-coded streams use the NSYM = 4 simple code (simple4=1|2 chooses its tree-select-0 form)
	if (opt.geti("fourvalues", 0)) for (Channel &c : ch) for (int y = 0; y < c.h; ++y) for (int x = 0; x < c.w; ++x) {
		const float r = Picture::hash01((uint64_t) x * 7919 + (uint64_t) y * 104729 + seed);
		c.at(x, y) = r < 0.55f ? 0 : r < 0.8f ? 1 : r < 0.93f ? 2 : 3;
	}
	simple4_mode() = opt.geti("simple4", 0);

	// ---- global transforms (coded order = forward order; the decoder undoes them last to first) ----
	std::vector<TransformW> transforms;
	if (palette) {
		// replace the colour channels by a palette (meta channel 0) + an index channel
		const int nb_colours = palette == 2 ? 24 : 40;
		const int nb_deltas = palette == 3 ? 6 : 0;
		Channel pal(nb_colours, 3), idx(W, H);
		for (int i = 0; i < nb_colours; ++i) for (int c = 0; c < 3; ++c) pal.at(i, c) = i < nb_deltas ? (int32_t) rng.below(9) - 4 : (int32_t) rng.below(256);
		for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
			int v = (ch[(size_t) colour0].at(x, y) * 3 + ch[(size_t) colour0 + 1].at(x, y) * 5 + ch[(size_t) colour0 + 2].at(x, y)) / 9;
			int i = (v * nb_colours) >> 8;
			if (palette == 2) { if (((x ^ y) & 31) == 5) i = -1 - (int) rng.below(140); else if (((x + 2 * y) & 63) == 9) i = nb_colours + (int) rng.below(64 + 125); }
			idx.at(x, y) = i;
		}
		std::vector<Channel> next;
		next.push_back(pal);
		next.push_back(idx);
		for (int c = 3; c < nch; ++c) next.push_back(ch[(size_t) c]);
		ch.swap(next);
		TransformW t; t.kind = 1; t.begin_c = colour0; t.num_c = 3; t.nb_colours = nb_colours; t.nb_deltas = nb_deltas; t.d_pred = palette == 3 ? 5 : 0;
		transforms.push_back(t);
	} else if (rct >= 0) {
		if (rct / 7 == 0 && rct % 7 != 2) forward_rct(ch, colour0, rct);   // other types: the picture is taken as already transformed
		TransformW t; t.kind = 0; t.begin_c = colour0; t.rct_type = rct;
		transforms.push_back(t);
	}
	const int nb_meta = palette ? 1 : 0;
	if (squeeze) {
		TransformW t; t.kind = 2;
		std::vector<SqueezeStep> steps = default_squeeze_steps(ch, nb_meta);
		if (squeeze == 3) {
			steps.clear();
			SqueezeStep a; a.horizontal = true; a.in_place = true; a.begin_c = 0; a.num_c = 3; steps.push_back(a);
			a.horizontal = false; steps.push_back(a);
			a.horizontal = true; a.in_place = false; a.begin_c = 1; a.num_c = 2; steps.push_back(a);
			a.horizontal = false; a.in_place = true; a.begin_c = 0; a.num_c = 1; steps.push_back(a);
			a.horizontal = false; a.in_place = false; a.begin_c = 0; a.num_c = (int) ch.size() >= 4 ? 4 : 3; steps.push_back(a);   // (with alpha: reaches into the residuals / the alpha channel)
		}
		if (squeeze >= 2) t.sq = steps;
		for (const SqueezeStep &st : steps) forward_squeeze_step(ch, st);
		transforms.push_back(t);
	}
	const int total_ch = (int) ch.size();

	// ---- MA tree ----
	MATree tree;
	{
		int root;
		if (tree_kind == 0) root = tree.leaf(5);
		else if (tree_kind == 4) root = tree.leaf(0);   // the zero predictor: residual = sample
		else if (tree_kind == 1) {
			int l0 = tree.leaf(5), l1 = tree.leaf(4), l2 = tree.leaf(13), l3 = tree.leaf(1), l4 = tree.leaf(2, 0, 0, 0), l5 = tree.leaf(12), l6 = tree.leaf(7), l7 = tree.leaf(3);
			int a = tree.branch(9, 100, l0, l1);      // W + N - NW
			int b = tree.branch(4, 40, l2, l3);       // |N|
			int c = tree.branch(12, -3, l4, l5);      // N - NE
			int d = tree.branch(3, 7, l6, l7);        // x
			int e = tree.branch(0, 1, a, b);          // channel index
			int f = tree.branch(2, 5, c, d);          // y
			root = tree.branch(10, 0, e, f);          // W - NW
		} else if (tree_kind == 2) {
			int l0 = tree.leaf(6), l1 = tree.leaf(6), l2 = tree.leaf(5), l3 = tree.leaf(6);
			int a = tree.branch(15, 8, l0, l1);       // max weighted-predictor error
			int b = tree.branch(15, -8, l2, l3);
			root = tree.branch(0, 1, a, b);
		} else if (tree_kind == 5) {
			// a wide tree: every property 0..14, every predictor but the weighted one, offsets and multipliers; 48 leaves
			static const int PREDS[13] = {5, 1, 2, 3, 4, 0, 7, 8, 9, 10, 11, 12, 13};
			static const int THR[15] = {1, 2, 60, 90, 30, 25, 110, 130, 2, 120, -1, 3, -4, 5, -2};
			uint32_t r = 12345u; int leaves = 0, splits = 0;
			std::function<int(int)> grow = [&](int depth) -> int {
				r = r * 1664525u + 1013904223u;
				if (depth == 0 || (depth < 4 && ((r >> 9) & 3) == 0)) {
					const int k = leaves++;
					return tree.leaf(PREDS[k % 13], (k % 5) - 2, k % 7 == 3 ? 1 : 0, k % 11 == 5 ? 2 : 0);
				}
				const int prop = splits++ % 15;
				const int thr = THR[prop] + (int) ((r >> 12) % 7) - 3;
				const int left = grow(depth - 1), right = grow(depth - 1);
				return tree.branch(prop, thr, left, right);
			};
			root = grow(6);
			while (leaves > 64) die("tree=5 grew past 64 leaves");
		} else {
			int l0 = tree.leaf(5), l1 = tree.leaf(5), l2 = tree.leaf(1), l3 = tree.leaf(2), l4 = tree.leaf(5);
			int a = tree.branch(16, 10, l0, l1);      // previous channel sample
			int b = tree.branch(19, 4, l2, l3);       // |previous channel sample - its gradient|
			int c = tree.branch(17, 60, a, b);        // |previous channel sample|
			root = tree.branch(0, 0, c, l4);          // channel 0 has no previous channel
		}
		tree.finalise(root);
	}
	CodeSpecW treespec; treespec.init(6, std::vector<uint8_t>(6, 0), 1); treespec.log_alpha = 6; treespec.cfg[0] = HybridCfg{4, 1, 0};
	StreamEncoder tree_enc(treespec); tree_tokens(tree, tree_enc); count_stream(treespec, tree_enc);

	auto make_spec = [&](const MATree &t) {
		CodeSpecW sp;
		const int nctx = t.num_ctx;
		std::vector<uint8_t> map((size_t) (nctx + (lz77 ? 1 : 0)));
		const int ncl = std::min(nctx, 3);
		for (int i = 0; i < nctx; ++i) map[(size_t) i] = (uint8_t) (i % ncl);
		int nclusters = ncl;
		if (lz77) { map[(size_t) nctx] = (uint8_t) ncl; nclusters = ncl + 1; }   // the distance context gets its own cluster
		sp.lz77 = lz77 != 0;
		sp.init(nctx, map, nclusters);
		sp.lz_min_symbol = 224; sp.lz_min_length = 3; sp.lz_len_cfg = HybridCfg{0, 0, 0};
		sp.use_prefix = use_prefix != 0; sp.log_alpha = 8;
		for (auto &c : sp.cfg) c = use_prefix ? HybridCfg{4, 2, 0} : HybridCfg{4, 1, 1};
		return sp;
	};
	CodeSpecW gspec = make_spec(tree);
	// localtree=1: every other group section repeats the global MA tree with its own code spec
	// (use_global_tree = 0, j40.h:3740); localtree=2: those sections use a different tree, one that
	// needs the weighted predictor whatever the global tree does
	const int local_tree = opt.geti("localtree", 0);
	MATree ltree = tree;
	if (local_tree == 2) {
		ltree = MATree();
		int l0 = ltree.leaf(6), l1 = ltree.leaf(5), l2 = ltree.leaf(4), l3 = ltree.leaf(1, 2, 0, 0);
		int a = ltree.branch(15, 8, l0, l1);      // max weighted-predictor error
		int b = ltree.branch(3, 100, l2, l3);     // x
		ltree.finalise(ltree.branch(2, 17, a, b)); // y
	}
	StreamEncoder ltree_enc(treespec);
	if (local_tree) { tree_tokens(ltree, ltree_enc); count_stream(treespec, ltree_enc); }
	const StreamEncoder ltree_saved = ltree_enc;   // flush() consumes the items
	const CodeSpecW lspec_proto = make_spec(ltree);
	WPParams wpp;
	const CodeSpecW &gspec_ref = gspec;

	// encodes the listed channels (sub-rectangles already cut out) into one stream; LZ77 replaces runs
	// of equal tokens ("previous symbol" = distance code 1 with a non-zero dist_mult, j40.h:2834)
	auto encode_image = [&](std::vector<Channel> &chs, int first, int64_t sidx, StreamEncoder &enc, bool local = false) {
		const MATree &use_tree = local ? ltree : tree;
		const CodeSpecW &gspec = local ? lspec_proto : gspec_ref;   // same cluster layout rules, the section's own contexts
		StreamEncoder raw(gspec);
		for (int c = first; c < (int) chs.size(); ++c) encode_channel(use_tree, chs, c, sidx, wpp, raw);
		if (!lz77) { enc.items = raw.items; return; }
		// run-length pass over the residual tokens: a run of >= 3 identical (cluster, token, extra) items
		// after its first occurrence becomes one copy with distance 1
		const auto &it = raw.items;
		for (size_t i = 0; i < it.size(); ) {
			size_t j = i + 1;
			while (j < it.size() && it[j].token == it[i].token && it[j].extra == it[i].extra && it[i].nextra == 0 && it[j].nextra == 0 && it[i].token < 16) ++j;
			enc.items.push_back(it[i]);
			size_t run = j - i - 1;
			if (run >= 3 && it[i].token < 16) {
				// the copied *values* are hybrid-decoded integers; with split_exp 4 a token < 16 is its own value
				uint32_t cl = it[i + 1].cluster;
				HToken t = hybrid_encode((uint32_t) run - (uint32_t) gspec.lz_min_length, gspec.lz_len_cfg);
				enc.items.push_back({cl, t.token + (uint32_t) gspec.lz_min_symbol, t.extra, (uint8_t) t.nextra});
				uint32_t lzcl = gspec.cluster_map[(size_t) gspec.total_dist() - 1];
				HToken d = hybrid_encode(1, gspec.cfg[lzcl]);   // special distance code 1 = previous symbol
				enc.items.push_back({lzcl, d.token, d.extra, (uint8_t) d.nextra});
				i = j;
			} else i = i + 1;
		}
	};

	std::vector<StreamEncoder> encs;
	std::vector<std::vector<TransformW>> group_tr;
	const int local_rct = opt.geti("localrct", -1);
	const int local_palette = opt.geti("localpalette", 0);
	std::vector<uint8_t> section_is_group;
	// squeeze, frames larger than a group: which encoder (index into encs) holds each LfGroup / pass-group section; -1 = nothing coded there
	std::vector<int> sq_lf_enc, sq_group_enc;
	if (single) {
		encs.emplace_back(gspec);
		encode_image(ch, 0, 0, encs.back());
	} else if (squeeze) {
		// The decoder deals the channels out by size and shift (ISO 18181-1): LfGlobal takes the meta channels and the channels behind
		// them that fit one group; an LfGroup section the channels shifted by >= 3 both ways, over its 8-groups-wide area; a pass-group
		// section the others over its own area. A channel's rectangle is the area shifted down, clipped to the channel.
		int num_gm = nb_meta;
		while (num_gm < total_ch && ch[(size_t) num_gm].w <= gdim && ch[(size_t) num_gm].h <= gdim) ++num_gm;
		encs.emplace_back(gspec);
		{ std::vector<Channel> head(ch.begin(), ch.begin() + num_gm); if (num_gm) encode_image(head, 0, 0, encs.back()); }
		std::map<uint64_t, int> seen;   // content hash of a section's channels -> its encoder
		auto cut = [&](bool lf, int left, int top, int dim, int64_t sidx) -> int {
			std::vector<Channel> sub;
			uint64_t hsh = 1469598103934665603ull;
			auto mix = [&](uint64_t v) { hsh = (hsh ^ v) * 1099511628211ull; };
			for (int c = num_gm; c < total_ch; ++c) {
				const Channel &full = ch[(size_t) c];
				if ((full.hshift >= 3 && full.vshift >= 3) != lf) continue;
				const int x0 = left >> full.hshift, y0 = top >> full.vshift, w = std::min(dim >> full.hshift, full.w - x0), h = std::min(dim >> full.vshift, full.h - y0);
				if (w <= 0 || h <= 0) continue;
				Channel s2(w, h); s2.hshift = full.hshift; s2.vshift = full.vshift;
				for (int y = 0; y < h; ++y) memcpy(&s2.at(0, y), full.px.data() + (size_t) (y0 + y) * (size_t) full.w + (size_t) x0, sizeof(int32_t) * (size_t) w);
				mix((uint64_t) w << 32 | (uint32_t) h); mix((uint64_t) full.hshift << 8 | (uint64_t) full.vshift);
				for (int32_t v : s2.px) mix((uint32_t) v);
				sub.push_back(std::move(s2));
			}
			if (sub.empty()) return -1;
			auto it = seen.find(hsh);
			if (it != seen.end()) return it->second;
			encs.emplace_back(gspec);
			encode_image(sub, 0, sidx, encs.back());
			seen[hsh] = (int) encs.size() - 1;
			return (int) encs.size() - 1;
		};
		const int lfdim = gdim * 8, lfcols = (W + lfdim - 1) / lfdim;
		for (int gg = 0; gg < num_lf_groups; ++gg) sq_lf_enc.push_back(cut(true, (gg % lfcols) * lfdim, (gg / lfcols) * lfdim, lfdim, 1 + num_lf_groups + gg));
		for (int g = 0; g < num_groups; ++g) sq_group_enc.push_back(cut(false, (g % gcols) * gdim, (g / gcols) * gdim, gdim, 1 + 3 * num_lf_groups + 17 + g));
	} else {
		// meta channels (palette) are decoded inside LfGlobal (num_gm_channels = nb_meta_channels, j40.h:6332)
		encs.emplace_back(gspec);
		if (nb_meta) { std::vector<Channel> meta(ch.begin(), ch.begin() + nb_meta); encode_image(meta, 0, 0, encs.back()); }
		// passes=P: every pass codes the whole group again (the reference decodes and pastes all channels in each pass,
		// j40.h:7025-7033, so the last pass is what stays); earlier passes carry a perturbed picture
		for (int pass = 0; pass < num_passes; ++pass) for (int g = 0; g < num_groups; ++g) {
			const int gx = (g % gcols) * gdim, gy = (g / gcols) * gdim, gw = std::min(gdim, W - gx), gh = std::min(gdim, H - gy);
			std::vector<Channel> sub;
			for (int c = nb_meta; c < total_ch; ++c) {
				Channel s2(gw, gh);
				for (int y = 0; y < gh; ++y) for (int x = 0; x < gw; ++x) s2.at(x, y) = ch[(size_t) c].at(gx + x, gy + y) ^ (pass + 1 < num_passes ? ((x + 2 * y + pass) & 3) : 0);
				sub.push_back(s2);
			}
			const bool local = local_tree && (((size_t) (pass * num_groups + g) + 1) & 1);
			// localrct=K: the group's header lists one or two RCTs of its own (j40.h:3757-3773), type varying by group
			group_tr.emplace_back();
			// localpalette=K (1 plain, 2 with deltas and synthetic colours, 3 with delta prediction): every other group replaces its
			// three colour channels by a palette of its own + an index channel (j40.h:3774-3799); those groups list no RCT
			const bool own_palette = local_palette && sub.size() >= 3 && ((g + pass) & 1) == 0;
			if (own_palette) {
				const int nb_colours = local_palette == 2 ? 20 : 33, nb_deltas = local_palette == 3 ? 5 : 0;
				Channel pal(nb_colours, 3), idx(gw, gh);
				for (int i = 0; i < nb_colours; ++i) for (int c = 0; c < 3; ++c) pal.at(i, c) = i < nb_deltas ? (int32_t) rng.below(9) - 4 : (int32_t) rng.below(256);
				for (int y = 0; y < gh; ++y) for (int x = 0; x < gw; ++x) {
					int v = (sub[0].at(x, y) * 3 + sub[1].at(x, y) * 5 + sub[2].at(x, y)) / 9;
					int i = ((v & 255) * nb_colours) >> 8;
					if (local_palette == 2) { if (((x ^ y) & 31) == 5) i = -1 - (int) rng.below(140); else if (((x + 2 * y) & 63) == 9) i = nb_colours + (int) rng.below(64 + 125); }
					idx.at(x, y) = i;
				}
				std::vector<Channel> next{pal, idx};
				for (size_t c = 3; c < sub.size(); ++c) next.push_back(sub[c]);
				sub.swap(next);
				TransformW t; t.kind = 1; t.begin_c = 0; t.num_c = 3; t.nb_colours = nb_colours; t.nb_deltas = nb_deltas; t.d_pred = local_palette == 3 ? 5 : 0;
				group_tr.back().push_back(t);
			}
			if (local_rct >= 0 && sub.size() >= 3 && !own_palette) {
				const int t1 = (local_rct + 5 * (g + pass)) % 42, t2 = (3 * t1 + 1) % 42;
				TransformW a; a.kind = 0; a.begin_c = 0; a.rct_type = t1; group_tr.back().push_back(a);
				if (t1 / 7 == 0 && t1 % 7 != 2) forward_rct(sub, 0, t1);
				if (g % 3 == 2) {
					TransformW b; b.kind = 0; b.begin_c = (int) sub.size() - 3; b.rct_type = t2; group_tr.back().push_back(b);
					if (t2 / 7 == 0 && t2 % 7 != 2) forward_rct(sub, b.begin_c, t2);
				}
			}
			encs.emplace_back(gspec);
			// stream index of a pass group (j40.h:7013): 1 + 3 * num_lf_groups + 17 + pass * num_groups + gidx
			encode_image(sub, 0, 1 + 3 * num_lf_groups + 17 + pass * num_groups + g, encs.back(), local);
		}
	}
	std::vector<CodeSpecW> lspec(encs.size(), lspec_proto);
	for (size_t i = 0; i < encs.size(); ++i) {
		if (local_tree && !single && i >= 1 && (i & 1)) { encs[i].spec = &lspec[i]; count_stream(lspec[i], encs[i]); }
		else count_stream(gspec, encs[i]);
	}

	// ---- sections ----
	std::vector<std::vector<uint8_t>> sections;
	{
		BitWriter bw;
		bw.put(1, 1);                 // LF channel dequantisation: default
		bw.put(1, 1);                 // global tree present
		write_code_spec(bw, treespec); tree_enc.flush(bw);
		write_code_spec(bw, gspec);
		write_modular_header(bw, true, nullptr, transforms);
		encs[0].flush(bw);
		bw.pad();
		sections.push_back(bw.bytes);
	}
	const int gcols_full = (Wfull + gdim - 1) / gdim, grows_full = (Hfull + gdim - 1) / gdim;
	const int num_lf_groups_full = ((Wfull + 8 * gdim - 1) / (8 * gdim)) * ((Hfull + 8 * gdim - 1) / (8 * gdim));
	if (!single && squeeze) {
		std::vector<std::vector<uint8_t>> done(encs.size());   // a shared encoder is serialised once
		auto section_of = [&](int e) -> std::vector<uint8_t> {
			if (e < 0) return {};
			if (done[(size_t) e].empty()) {
				BitWriter bw;
				write_modular_header(bw, true, nullptr, {});
				encs[(size_t) e].flush(bw);
				bw.pad();
				done[(size_t) e] = bw.bytes;
			}
			return done[(size_t) e];
		};
		for (int e : sq_lf_enc) sections.push_back(section_of(e));
		sections.push_back({});       // HfGlobal: empty in Modular frames
		for (int e : sq_group_enc) sections.push_back(section_of(e));
	} else if (!single) {
		for (int i = 0; i < num_lf_groups_full; ++i) sections.push_back({});
		sections.push_back({});       // HfGlobal must be empty for Modular frames (j40.h:7825)
		for (int g = 0; g < num_passes * num_groups; ++g) {
			BitWriter bw;
			if (local_tree && (((size_t) g + 1) & 1)) {
				write_modular_header(bw, false, nullptr, group_tr[(size_t) g]);
				StreamEncoder te = ltree_saved;
				write_code_spec(bw, treespec); te.flush(bw);
				write_code_spec(bw, lspec[(size_t) g + 1]);
			} else write_modular_header(bw, true, nullptr, group_tr[(size_t) g]);
			encs[(size_t) g + 1].flush(bw);
			bw.pad();
			sections.push_back(bw.bytes);
		}
		if (repeat > 1) {   // lay the base picture's group sections out over the full frame
			const size_t first = sections.size() - (size_t) num_groups;
			std::vector<std::vector<uint8_t>> base(sections.begin() + (long) first, sections.end());
			sections.resize(first);
			for (int gy = 0; gy < grows_full; ++gy) for (int gx = 0; gx < gcols_full; ++gx) sections.push_back(base[(size_t) (gy % grows) * (size_t) gcols + (size_t) (gx % gcols)]);
		}
	}

	// ---- codestream ----
	BitWriter cs;
	cs.put(0xff, 8); cs.put(0x0a, 8);
	write_size_header(cs, Wfull, Hfull);
	cs.put(0, 1);                       // ImageMetadata: not all_default
	cs.put(0, 1);                       // no extra fields
	write_bit_depth(cs, bpp);
	cs.put(1, 1);                       // modular_16bit_buffers
	{   // num_extra_channels U32(0, 1, 2 + u(4), 1 + u(12)), then one ExtraChannelInfo each (j40.h:3247-3290)
		const int nec = extra + (alpha ? 1 : 0);
		if (nec == 0) cs.put(0, 2); else if (nec == 1) cs.put(1, 2); else { cs.put(2, 2); cs.put((uint64_t) (nec - 2), 4); }
		for (int k = 0; k < extra; ++k) {
			cs.put(0, 1);                   // not the default alpha
			cs.put(1, 2);                   // type: enum selector 1 = depth
			write_bit_depth(cs, bpp);
			cs.put(0, 2);                   // dim_shift 0
			cs.put(0, 2);                   // no name
		}
		if (alpha) cs.put(1, 1);            // d_alpha
	}
	// xyb=1 / ycbcr=1: the frame is flagged XYB / YCbCr; the reference applies no colour transform to Modular
	// frames (j40.h:8209-8210, 7910) and renders the three channels as they are
	const int flag_xyb = opt.geti("xyb", 0), flag_ycbcr = opt.geti("ycbcr", 0);
	cs.put(flag_xyb ? 1 : 0, 1);        // xyb_encoded
	const int icc_bytes = opt.geti("icc", 0);
	if (!icc_bytes) cs.put(1, 1);       // ColourEncoding.all_default (sRGB)
	else { cs.put(0, 1); cs.put(1, 1); cs.put(0, 2); }   // want_icc, colour_space = RGB
	cs.put(0, 2);                       // extensions
	cs.put(1, 1);                       // default_m
	if (icc_bytes) write_icc_stream(cs, rng, icc_bytes);
	cs.pad();
	cs.put(0, 1);                       // FrameHeader: not all_default
	cs.put(0, 2);                       // regular frame
	cs.put(1, 1);                       // modular
	cs.u64(0);                          // flags
	if (!flag_xyb) cs.put(flag_ycbcr ? 1 : 0, 1);   // do_ycbcr
	if (!flag_xyb && flag_ycbcr) cs.put(0, 6);      // jpeg_upsampling: none
	cs.put(0, 2);                       // log_upsampling
	for (int k = 0; k < extra + (alpha ? 1 : 0); ++k) cs.put(0, 2);   // extra channel upsampling
	cs.put((uint64_t) (group_shift - 7), 2);
	cs.u32(num_passes, 1, 0, 2, 0, 3, 0, 4, 3);
	if (num_passes > 1) {
		cs.u32(0, 0, 0, 1, 0, 2, 0, 3, 1);                  // num_ds = 0
		for (int i = 0; i < num_passes - 1; ++i) cs.put(0, 2);  // shift[i]
	}
	cs.put(0, 1);                       // have_crop
	cs.u32(0, 0, 0, 1, 0, 2, 0, 3, 2);  // blend mode (colour)
	for (int k = 0; k < extra + (alpha ? 1 : 0); ++k) cs.u32(0, 0, 0, 1, 0, 2, 0, 3, 2);  // blend mode (extra channels)
	cs.put(1, 1);                       // is_last
	cs.u32(0, 0, 0, 0, 4, 16, 5, 48, 10);
	cs.put(0, 1); cs.put(0, 1); cs.put(0, 2); cs.u64(0);   // restoration: explicit, gab off, epf 0, no extensions
	cs.u64(0);                          // frame extensions
	write_toc_and_sections(cs, sections, opt.geti("permute", 0), rng, opt.geti("slack", 0));
	std::vector<uint8_t> file = cs.bytes;
	if (container) {
		static const uint8_t HEAD[32] = {0, 0, 0, 0x0c, 'J', 'X', 'L', ' ', 0x0d, 0x0a, 0x87, 0x0a, 0, 0, 0, 0x14, 'f', 't', 'y', 'p', 'j', 'x', 'l', ' ', 0, 0, 0, 0, 'j', 'x', 'l', ' '};
		file.assign(HEAD, HEAD + 32);
		size_t total = 8 + cs.bytes.size();
		uint8_t hd[8] = {(uint8_t) (total >> 24), (uint8_t) (total >> 16), (uint8_t) (total >> 8), (uint8_t) total, 'j', 'x', 'l', 'c'};
		file.insert(file.end(), hd, hd + 8);
		file.insert(file.end(), cs.bytes.begin(), cs.bytes.end());
	}
	if (!write_file(out, file)) die("cannot write output");
	fprintf(stderr, "modular %dx%d: %zu bytes (%.3f bpp), %d groups%s\n", Wfull, Hfull, file.size(), 8.0 * (double) file.size() / ((double) Wfull * Hfull), gcols_full * grows_full, single ? " (single section)" : "");
	return 0;
}


int main(int argc, char **argv) {
	if (argc < 6) { fprintf(stderr, "usage: %s vardct|modular W H SEED OUT [key=value ...]\n", argv[0]); return 1; }
	Options opt;
	for (int i = 6; i < argc; ++i) { std::string a = argv[i]; size_t eq = a.find('='); if (eq == std::string::npos) die("options are key=value"); opt.kv[a.substr(0, eq)] = a.substr(eq + 1); }
	int W = atoi(argv[2]), H = atoi(argv[3]); uint64_t seed = strtoull(argv[4], nullptr, 0);
	if (W <= 0 || H <= 0) die("bad dimensions");
	if (!strcmp(argv[1], "vardct")) return run_vardct(W, H, seed, argv[5], opt);
	if (!strcmp(argv[1], "modular")) return run_modular(W, H, seed, argv[5], opt);
	die("unknown mode");
}
