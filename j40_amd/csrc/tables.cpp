// j40_amd/csrc/tables.cpp -- host-built tables the hot path reads: transform descriptors, default
// dequantisation weights, coefficient orders, DCT constants, and the LF -> LLF forward transform.
//
// The numeric tables must equal the *float32 values the reference uses*, not the mathematically
// best ones: HALF_SECANTS / LF2LLF_SCALES are regenerated from their
// formulas with the reference's decimal rounding (and pinned against oracle/_ref in tests);
// dequantisation weights are evaluated with libm in the reference's operation order (j40.h:4780-4957).
#include "frame.hpp"
#include "tables.hpp"
#include "device/plan.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace j40hip {

// DctSelect -> (log rows, log columns, dequant parameter set, coefficient order); spec table
// (cf. j40.h:4591)
const DctSelect DCT_SELECT[27] = {
	{3, 3, 0, 0}, {3, 3, 1, 1}, {3, 3, 2, 1}, {3, 3, 3, 1}, {4, 4, 4, 2}, {5, 5, 5, 3}, {4, 3, 6, 4}, {3, 4, 6, 4}, {5, 3, 7, 5},
	{3, 5, 7, 5}, {5, 4, 8, 6}, {4, 5, 8, 6}, {3, 3, 9, 1}, {3, 3, 9, 1}, {3, 3, 10, 1}, {3, 3, 10, 1}, {3, 3, 10, 1}, {3, 3, 10, 1},
	{6, 6, 11, 7}, {6, 5, 12, 8}, {5, 6, 12, 8}, {7, 7, 13, 9}, {7, 6, 14, 10}, {6, 7, 14, 10}, {8, 8, 15, 11}, {8, 7, 16, 12}, {7, 8, 16, 12},
};
const int8_t LOG_ORDER_SIZE[13][2] = {{3, 3}, {3, 3}, {4, 4}, {5, 5}, {3, 4}, {3, 5}, {4, 5}, {6, 6}, {5, 6}, {7, 7}, {6, 7}, {8, 8}, {7, 8}};

// per dequant parameter set: log rows / columns of the weight matrix (short side first)
static const int8_t DQ_LOG_DIMS[17][2] = {{3, 3}, {3, 3}, {3, 3}, {3, 3}, {4, 4}, {5, 5}, {3, 4}, {3, 5}, {4, 5}, {3, 3}, {3, 3}, {6, 6}, {5, 6}, {7, 7}, {6, 7}, {8, 8}, {7, 8}};

// ------------------------------------------------------------------------------------------------
// DCT constants

static float g_half_secants[256], g_lf2llf[64];
static std::once_flag g_tables_once;

static void init_float_tables() {
	const double PI = 3.14159265358979323846;
	char buf[64];
	g_half_secants[0] = g_half_secants[1] = 0.0f;
	for (int n = 1; n <= 7; ++n) for (int k = 0; k < (1 << n); ++k) {
		double v = 1.0 / (2.0 * cos(((double) k + 0.5) * PI / (double) (1 << (n + 1))));
		snprintf(buf, sizeof buf, v < 10.0 ? "%.8f" : "%.7f", v);   // the reference's literals are decimal print-outs
		g_half_secants[(1 << n) + k] = strtof(buf, nullptr);
	}
	g_lf2llf[0] = 0.0f;
	for (int N = 1; N <= 32; N *= 2) for (int k = 0; k < N; ++k) {
		double v = 1.0 / (cos((double) k * PI / (16.0 * N)) * cos((double) k * PI / (8.0 * N)) * cos((double) k * PI / (4.0 * N)) * (double) N);
		snprintf(buf, sizeof buf, "%.8f", v);
		g_lf2llf[N + k] = strtof(buf, nullptr);
	}
}

const float *half_secants() { std::call_once(g_tables_once, init_float_tables); return g_half_secants; }
const float *lf2llf_scales() { std::call_once(g_tables_once, init_float_tables); return g_lf2llf; }

// AFV basis (ISO 18181-1 AFVBasis, stored basis-major: AFV_BASIS[sample * 16 + coefficient]),
// eight-decimal spec constants
static const float AFV_BASIS_TABLE[256] = {
	0.25000000f, 0.87690293f, 0.00000000f, 0.00000000f, 0.00000000f, -0.41053776f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f,
	0.25000000f, 0.22065181f, 0.00000000f, 0.00000000f, -0.70710678f, 0.62354854f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f,
	0.25000000f, -0.10140050f, 0.40670076f, -0.21255748f, 0.00000000f, -0.06435072f, -0.45175566f, -0.30468475f, 0.30179295f, 0.40824829f, 0.17478670f, -0.21105601f, -0.14266085f, -0.13813540f, -0.17437603f, 0.11354987f,
	0.25000000f, -0.10140050f, 0.44444817f, 0.30854971f, 0.00000000f, -0.06435072f, 0.15854504f, 0.51126161f, 0.25792363f, 0.00000000f, 0.08126112f, 0.18567181f, -0.34164468f, 0.33022826f, 0.07027907f, -0.07417505f,
	0.25000000f, 0.22065181f, 0.00000000f, 0.00000000f, 0.70710678f, 0.62354854f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f, 0.00000000f,
	0.25000000f, -0.10140050f, 0.00000000f, 0.47067023f, 0.00000000f, -0.06435072f, -0.04038515f, 0.00000000f, 0.16272340f, 0.00000000f, 0.00000000f, 0.00000000f, 0.73674975f, 0.08755115f, -0.29210266f, 0.19402893f,
	0.25000000f, -0.10140050f, 0.19574399f, -0.16212052f, 0.00000000f, -0.06435072f, 0.00741823f, -0.29048013f, 0.09520023f, 0.00000000f, -0.36753980f, 0.49215859f, 0.24627108f, -0.07946707f, 0.36238173f, -0.43519050f,
	0.25000000f, -0.10140050f, 0.29291001f, 0.00000000f, 0.00000000f, -0.06435072f, 0.39351034f, -0.06578702f, 0.00000000f, -0.40824829f, -0.30788221f, -0.38525014f, -0.08574019f, -0.46133749f, 0.00000000f, 0.21918685f,
	0.25000000f, -0.10140050f, -0.40670076f, -0.21255748f, 0.00000000f, -0.06435072f, -0.45175566f, 0.30468475f, 0.30179295f, -0.40824829f, -0.17478670f, 0.21105601f, -0.14266085f, -0.13813540f, -0.17437603f, 0.11354987f,
	0.25000000f, -0.10140050f, -0.19574399f, -0.16212052f, 0.00000000f, -0.06435072f, 0.00741823f, 0.29048013f, 0.09520023f, 0.00000000f, 0.36753980f, -0.49215859f, 0.24627108f, -0.07946707f, 0.36238173f, -0.43519050f,
	0.25000000f, -0.10140050f, 0.00000000f, -0.47067023f, 0.00000000f, -0.06435072f, 0.11074166f, 0.00000000f, -0.16272340f, 0.00000000f, 0.00000000f, 0.00000000f, 0.14883399f, 0.49724647f, 0.29210266f, 0.55504438f,
	0.25000000f, -0.10140050f, 0.11379074f, -0.14642919f, 0.00000000f, -0.06435072f, 0.08298163f, -0.23889774f, -0.35312385f, -0.40824829f, 0.48266891f, 0.17419413f, -0.04768680f, 0.12538059f, -0.43266080f, -0.25468277f,
	0.25000000f, -0.10140050f, -0.44444817f, 0.30854971f, 0.00000000f, -0.06435072f, 0.15854504f, -0.51126161f, 0.25792363f, 0.00000000f, -0.08126112f, -0.18567181f, -0.34164468f, 0.33022826f, 0.07027907f, -0.07417505f,
	0.25000000f, -0.10140050f, -0.29291001f, 0.00000000f, 0.00000000f, -0.06435072f, 0.39351034f, 0.06578702f, 0.00000000f, 0.40824829f, 0.30788221f, 0.38525014f, -0.08574019f, -0.46133749f, 0.00000000f, 0.21918685f,
	0.25000000f, -0.10140050f, -0.11379074f, -0.14642919f, 0.00000000f, -0.06435072f, 0.08298163f, 0.23889774f, -0.35312385f, 0.40824829f, -0.48266891f, -0.17419413f, -0.04768680f, 0.12538059f, -0.43266080f, -0.25468277f,
	0.25000000f, -0.10140050f, 0.00000000f, 0.42511496f, 0.00000000f, -0.06435072f, -0.45175566f, 0.00000000f, -0.60358590f, 0.00000000f, 0.00000000f, 0.00000000f, -0.14266085f, -0.13813540f, 0.34875205f, 0.11354987f,
};
const float *afv_basis() { return AFV_BASIS_TABLE; }

// thr[k] = smallest float v in (-9, 50000) whose 8-bit sample (j40.h:7213-7240 then 7925-7935 with bpp = 8) is >= k
const float *srgb_u8_thresholds() {
	static float table[SRGB_TABLE_FLOATS];
	static std::once_flag once;
	std::call_once(once, []() {
		auto sample = [](float v) -> int {   // the reference's arithmetic with a correctly rounded powf
			float t = v <= 0.0031308f ? 12.92f * v : 1.055f * (float) pow((double) v, (double) (1.0f / 2.4f)) - 0.055f;
			const float y = 255.0f * t + 0.5f;
			int32_t i = !(y > -2147483904.0f && y < 2147483648.0f) ? (int32_t) 0x80000000u : (int32_t) y;
			i = (int32_t) (int16_t) (uint16_t) (uint32_t) i;
			return i < 0 ? 0 : i > 255 ? 255 : i;
		};
		// positive floats order like their bit patterns; every threshold is positive (sample(0) = 0)
		auto from_bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
		uint32_t top; { const float t = 50000.0f; memcpy(&top, &t, 4); }
		table[0] = -INFINITY; table[256] = table[257] = INFINITY;
		for (int k = 1; k <= 255; ++k) {
			uint32_t lo = 0, hi = top;   // sample(lo) < k <= sample(hi)
			while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (sample(from_bits(mid)) >= k) hi = mid; else lo = mid; }
			table[k] = from_bits(hi);
		}
		// the buckets: sample at the bucket's smallest float (bucket 0 also takes everything below 2^-13, the last one everything
		// from 1.0 up); a bucket must not span more than one step for the device's single comparison to settle the sample
		uint8_t *bucket = (uint8_t *) (table + SRGB_THRESHOLDS);
		memset(bucket, 0, SRGB_BUCKETS);
		for (int32_t b = 0; b <= SRGB_BUCKET_HI - SRGB_BUCKET_LO; ++b) {
			const float lo = from_bits((uint32_t) (b + SRGB_BUCKET_LO) << 16);
			int k = 0; while (k < 255 && table[k + 1] <= lo) ++k;
			bucket[b] = (uint8_t) (b == 0 ? 0 : k);
			if (b > 0 && (int) bucket[b] - (int) bucket[b - 1] > 1) abort();   // cannot happen: the steepest bucket spans 0.66 of a step
		}
	});
	return table;
}

// ------------------------------------------------------------------------------------------------
// default ("library") dequantisation parameters, ISO 18181-1 with the corrections the reference
// applies (j40.h:4630-4690). Stored per parameter set as rows of {X, Y, B}.

typedef float f3[3];
#define DCT4X4_BANDS {2200.0f, 392.0f, 112.0f}, {0.0f, 0.0f, -0.25f}, {0.0f, 0.0f, -0.25f}, {0.0f, 0.0f, -0.5f}
#define DCT4X8_BANDS {2198.050556016380522f, 764.3655248643528689f, 527.107573587542228f}, {-0.96269623020744692f, -0.92630200888366945f, -1.4594385811273854f}, \
	{-0.76194253026666783f, -0.9675229603596517f, -1.450082094097871593f}, {-0.6551140670773547f, -0.27845290869168118f, -1.5843722511996204f}
#define LARGE_BANDS(k) {k * 23629.073922049845f, k * 8611.3238710010046f, k * 4492.2486445538634f}, {-1.025f, -0.3041958212306401f, -1.2f}, \
	{-0.78f, 0.3633036457487539f, -1.2f}, {-0.65012f, -0.35660379990111464f, -0.8f}, {-0.19041574084286472f, -0.3443074455424403f, -0.7f}, \
	{-0.20819395464f, -0.33699592683512467f, -0.7f}, {-0.421064f, -0.30180866526242109f, -0.4f}, {-0.32733845535848671f, -0.27321683125358037f, -0.5f}

static const f3 LIB_DCT8[] = {{3150.0f, 560.0f, 512.0f}, {0.0f, 0.0f, -2.0f}, {-0.4f, -0.3f, -1.0f}, {-0.4f, -0.3f, 0.0f}, {-0.4f, -0.3f, -1.0f}, {-2.0f, -0.3f, -2.0f}};
static const f3 LIB_HORNUSS[] = {{280.0f, 60.0f, 18.0f}, {3160.0f, 864.0f, 200.0f}, {3160.0f, 864.0f, 200.0f}};
static const f3 LIB_DCT2[] = {{3840.0f, 960.0f, 640.0f}, {2560.0f, 640.0f, 320.0f}, {1280.0f, 320.0f, 128.0f}, {640.0f, 180.0f, 64.0f}, {480.0f, 140.0f, 32.0f}, {300.0f, 120.0f, 16.0f}};
static const f3 LIB_DCT4[] = {{1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f}, DCT4X4_BANDS};
static const f3 LIB_DCT16[] = {
	{8996.8725711814115328f, 3191.48366296844234752f, 1157.50408145487200256f}, {-1.3000777393353804f, -0.67424582104194355f, -2.0531423165804414f},
	{-0.49424529824571225f, -0.80745813428471001f, -1.4f}, {-0.439093774457103443f, -0.44925837484843441f, -0.50687130033378396f},
	{-0.6350101832695744f, -0.35865440981033403f, -0.42708730624733904f}, {-0.90177264050827612f, -0.31322389111877305f, -1.4856834539296244f},
	{-1.6162099239887414f, -0.37615025315725483f, -4.9209142884401604f}};
static const f3 LIB_DCT32[] = {
	{15718.40830982518931456f, 7305.7636810695983104f, 3803.53173721215041536f}, {-1.025f, -0.8041958212306401f, -3.060733579805728f},
	{-0.98f, -0.7633036457487539f, -2.0413270132490346f}, {-0.9012f, -0.55660379990111464f, -2.0235650159727417f},
	{-0.4f, -0.49785304658857626f, -0.5495389509954993f}, {-0.48819395464f, -0.43699592683512467f, -0.4f},
	{-0.421064f, -0.40180866526242109f, -0.4f}, {-0.27f, -0.27321683125358037f, -0.3f}};
static const f3 LIB_DCT8X16[] = {
	{7240.7734393502f, 1448.15468787004f, 506.854140754517f}, {-0.7f, -0.5f, -1.4f}, {-0.7f, -0.5f, -0.2f}, {-0.2f, -0.5f, -0.5f},
	{-0.2f, -0.2f, -0.5f}, {-0.2f, -0.2f, -1.5f}, {-0.5f, -0.2f, -3.6f}};
static const f3 LIB_DCT8X32[] = {
	{16283.2494710648897f, 5089.15750884921511936f, 3397.77603275308720128f}, {-1.7812845336559429f, -0.320049391452786891f, -0.321327362693153371f},
	{-1.6309059012653515f, -0.35362849922161446f, -0.34507619223117997f}, {-1.0382179034313539f, -0.30340000000000003f, -0.70340000000000003f},
	{-0.85f, -0.61f, -0.9f}, {-0.7f, -0.5f, -1.0f}, {-0.9f, -0.5f, -1.0f}, {-1.2360638576849587f, -0.6f, -1.1754605576265209f}};
static const f3 LIB_DCT16X32[] = {
	{13844.97076442300573f, 4798.964084220744293f, 1807.236946760964614f}, {-0.97113799999999995f, -0.61125308982767057f, -1.2f},
	{-0.658f, -0.83770786552491361f, -1.2f}, {-0.42026f, -0.79014862079498627f, -0.7f}, {-0.22712f, -0.2692727459704829f, -0.7f},
	{-0.2206f, -0.38272769465388551f, -0.7f}, {-0.226f, -0.22924222653091453f, -0.4f}, {-0.6f, -0.20719098826199578f, -0.5f}};
static const f3 LIB_DCT4X8[] = {{1.0f, 1.0f, 1.0f}, DCT4X8_BANDS};
static const f3 LIB_AFV[] = {
	{3072.0f, 1024.0f, 384.0f}, {3072.0f, 1024.0f, 384.0f}, {256.0f, 50.0f, 12.0f}, {256.0f, 50.0f, 12.0f}, {256.0f, 50.0f, 12.0f}, {414.0f, 58.0f, 22.0f},
	{0.0f, 0.0f, -0.25f}, {0.0f, 0.0f, -0.25f}, {0.0f, 0.0f, -0.25f}, DCT4X8_BANDS, DCT4X4_BANDS};
static const f3 LIB_DCT64[] = {LARGE_BANDS(0.9f)};
static const f3 LIB_DCT32X64[] = {LARGE_BANDS(0.65f)};
static const f3 LIB_DCT128[] = {LARGE_BANDS(1.8f)};
static const f3 LIB_DCT64X128[] = {LARGE_BANDS(1.3f)};
static const f3 LIB_DCT256[] = {LARGE_BANDS(3.6f)};
static const f3 LIB_DCT128X256[] = {LARGE_BANDS(2.6f)};

enum { ENC_LIBRARY = 0, ENC_HORNUSS = 1, ENC_DCT2 = 2, ENC_DCT4 = 3, ENC_DCT4X8 = 4, ENC_AFV = 5, ENC_DCT = 6, ENC_RAW = 7 };
static const struct { const f3 *params; int8_t mode, n, m; } LIBRARY[17] = {
	{LIB_DCT8, ENC_DCT, 6, 0}, {LIB_HORNUSS, ENC_HORNUSS, 0, 0}, {LIB_DCT2, ENC_DCT2, 0, 0}, {LIB_DCT4, ENC_DCT4, 4, 0},
	{LIB_DCT16, ENC_DCT, 7, 0}, {LIB_DCT32, ENC_DCT, 8, 0}, {LIB_DCT8X16, ENC_DCT, 7, 0}, {LIB_DCT8X32, ENC_DCT, 8, 0},
	{LIB_DCT16X32, ENC_DCT, 8, 0}, {LIB_DCT4X8, ENC_DCT4X8, 4, 0}, {LIB_AFV, ENC_AFV, 4, 4}, {LIB_DCT64, ENC_DCT, 8, 0},
	{LIB_DCT32X64, ENC_DCT, 8, 0}, {LIB_DCT128, ENC_DCT, 8, 0}, {LIB_DCT64X128, ENC_DCT, 8, 0}, {LIB_DCT256, ENC_DCT, 8, 0},
	{LIB_DCT128X256, ENC_DCT, 8, 0},
};

typedef std::array<float, 3> w3;

static void interpolation_bands(const w3 *params, int32_t n, w3 *out) {  // j40.h:4792
	for (int c = 0; c < 3; ++c) {
		out[0][(size_t) c] = params[0][(size_t) c];
		J40HIP_SHOULD(out[0][(size_t) c] > 0, "band");
		for (int32_t i = 1; i < n; ++i) {
			float v = params[i][(size_t) c];
			out[i][(size_t) c] = v > 0 ? out[i - 1][(size_t) c] * (1.0f + v) : out[i - 1][(size_t) c] / (1.0f - v);
			J40HIP_SHOULD(out[i][(size_t) c] > 0, "band");
		}
	}
}

static float interpolate(float pos, int c, const w3 *bands, int32_t len) {  // j40.h:4780
	if (len == 1) return bands[0][(size_t) c];
	float scaled_pos = pos * (float) (len - 1);
	int32_t scaled_idx = (int32_t) scaled_pos;
	float frac_idx = scaled_pos - (float) scaled_idx;
	float a = bands[scaled_idx][(size_t) c], b = bands[scaled_idx + 1][(size_t) c];
	return a * powf(b / a, frac_idx);
}

static void dct_quant_weights(int32_t rows, int32_t columns, const w3 *bands, int32_t len, w3 *out) {  // j40.h:4811
	const float inv_rows_m1 = 1.0f / (float) (rows - 1), inv_columns_m1 = 1.0f / (float) (columns - 1);
	const float INV_SQRT2 = 1.0f / 1.414214562373095f;
	for (int c = 0; c < 3; ++c) for (int32_t y = 0; y < rows; ++y) for (int32_t x = 0; x < columns; ++x) {
		float d = hypotf((float) x * inv_columns_m1, (float) y * inv_rows_m1);
		out[y * columns + x][(size_t) c] = interpolate(d * INV_SQRT2, c, bands, len);
	}
}

void load_dq_matrix(int32_t idx, DqMatrix *dq) {  // j40.h:4828
	if (dq->loaded) return;
	if (dq->mode == ENC_RAW) { dq->loaded = true; return; }
	int32_t mode = dq->mode, n = dq->n, m = dq->m;
	std::vector<w3> params_store;
	const w3 *params;
	if (mode == ENC_LIBRARY) {
		mode = LIBRARY[idx].mode; n = LIBRARY[idx].n; m = LIBRARY[idx].m;
		int32_t count = mode == ENC_DCT ? n : mode == ENC_HORNUSS ? 3 : mode == ENC_DCT2 ? 6 : mode == ENC_DCT4 ? 2 + n : mode == ENC_DCT4X8 ? 1 + n : 9 + n + m;
		params_store.resize((size_t) count);
		for (int32_t i = 0; i < count; ++i) for (int c = 0; c < 3; ++c) params_store[(size_t) i][(size_t) c] = LIBRARY[idx].params[i][c];
		params = params_store.data();
	} else {
		params = dq->params.data();
	}
	const int32_t rows = 1 << DQ_LOG_DIMS[idx][0], columns = 1 << DQ_LOG_DIMS[idx][1];
	std::vector<w3> raw((size_t) (rows * columns));
	w3 bands[15], scratch[64];
	switch (mode) {
	case ENC_DCT:
		interpolation_bands(params, n, bands);
		dct_quant_weights(rows, columns, bands, n, raw.data());
		break;
	case ENC_DCT4:
		interpolation_bands(params + 2, n, bands);
		dct_quant_weights(4, 4, bands, n, scratch);
		for (int c = 0; c < 3; ++c) {
			for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) raw[(size_t) (y * 8 + x)][(size_t) c] = scratch[(y / 2) * 4 + (x / 2)][(size_t) c];
			raw[1][(size_t) c] /= params[0][(size_t) c];
			raw[8][(size_t) c] /= params[0][(size_t) c];
			raw[9][(size_t) c] /= params[1][(size_t) c];
		}
		break;
	case ENC_DCT2:
		for (int c = 0; c < 3; ++c) {
			// band index per coefficient: rings around the top-left corner
			for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) {
				int r = x > y ? x : y, band;
				if (r >= 4) band = (x >= 4 && y >= 4) ? 5 : 4;
				else if (r >= 2) band = (x >= 2 && y >= 2) ? 3 : 2;
				else band = (x == 1 && y == 1) ? 1 : 0;
				raw[(size_t) (y * 8 + x)][(size_t) c] = params[band][(size_t) c];
			}
			raw[0][(size_t) c] = -1.0f;
		}
		break;
	case ENC_HORNUSS:
		for (int c = 0; c < 3; ++c) {
			for (int i = 0; i < 64; ++i) raw[(size_t) i][(size_t) c] = params[0][(size_t) c];
			raw[0][(size_t) c] = 1.0f;
			raw[1][(size_t) c] = raw[8][(size_t) c] = params[1][(size_t) c];
			raw[9][(size_t) c] = params[2][(size_t) c];
		}
		break;
	case ENC_DCT4X8:
		interpolation_bands(params + 1, n, bands);
		dct_quant_weights(4, 8, bands, n, scratch);
		for (int c = 0; c < 3; ++c) {
			for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) raw[(size_t) (y * 8 + x)][(size_t) c] = scratch[(y / 2) * 8 + x][(size_t) c];
			raw[1][(size_t) c] /= params[0][(size_t) c];
		}
		break;
	case ENC_AFV: {
		interpolation_bands(params + 9, n, bands);
		dct_quant_weights(4, 8, bands, n, scratch);
		interpolation_bands(params + 9 + n, m, bands);
		dct_quant_weights(4, 4, bands, m, scratch + 32);
		interpolation_bands(params + 5, 4, bands);
		// (freqs[i] - lo) / (hi - lo + 1e-6) for the twelve AFV frequencies (j40.h:4931)
		static const float FREQS[12] = {0.000000000f, 0.373436417f, 0.320380100f, 0.379332596f, 0.066671353f, 0.259756761f, 0.530035651f, 0.789731061f, 0.149436598f, 0.559318823f, 0.669198646f, 0.999999917f};
		for (int c = 0; c < 3; ++c) {
			scratch[0][(size_t) c] = params[0][(size_t) c];
			scratch[32][(size_t) c] = params[1][(size_t) c];
			for (int i = 0; i < 12; ++i) scratch[i + 48][(size_t) c] = interpolate(FREQS[i], c, bands, 4);
			scratch[60][(size_t) c] = 1.0f;
			for (int i = 0; i < 3; ++i) scratch[i + 61][(size_t) c] = params[i + 2][(size_t) c];
		}
		// odd rows come from the 4x8 weights; even rows interleave the 4x4 weights (odd columns) with
		// the AFV weights (even columns), whose corner cells are direct parameters (j40.h:4943-4955)
		{
			static const int8_t AFV_WEIGHT_SOURCE[64] = {
				60, 32, 62, 33, 48, 34, 49, 35, 0, 1, 2, 3, 4, 5, 6, 7, 61, 36, 63, 37, 50, 38, 51, 39, 8, 9, 10, 11, 12, 13, 14, 15,
				52, 40, 53, 41, 54, 42, 55, 43, 16, 17, 18, 19, 20, 21, 22, 23, 56, 44, 57, 45, 58, 46, 59, 47, 24, 25, 26, 27, 28, 29, 30, 31,
			};
			for (int c = 0; c < 3; ++c) for (int i = 0; i < 64; ++i) raw[(size_t) i][(size_t) c] = scratch[AFV_WEIGHT_SOURCE[i]][(size_t) c];
		}
		break;
	}
	default: J40HIP_RAISE("dqm?");
	}
	dq->params.swap(raw);
	dq->mode = ENC_RAW; dq->n = rows; dq->m = columns; dq->loaded = true;
}

// ------------------------------------------------------------------------------------------------
// natural coefficient order for a (1 << log_rows) x (1 << log_columns) block with log_columns >=
// log_rows: the LLF corner first (row-major), then anti-diagonals of the block stretched to a
// square, alternating direction (ISO 18181-1 natural ordering; cf. j40.h:4980)

void natural_order(int32_t log_rows, int32_t log_columns, std::vector<int32_t> *out) {
	const int32_t rows = 1 << log_rows, columns = 1 << log_columns, slope = 1 << (log_columns - log_rows);
	const int32_t rows8 = rows >> 3, columns8 = columns >> 3;
	std::vector<int32_t> order;
	order.reserve((size_t) rows * (size_t) columns);
	for (int32_t y = 0; y < rows8; ++y) for (int32_t x = 0; x < columns8; ++x) order.push_back(y << log_columns | x);
	// a diagonal is the set of cells with x + y * slope == key and x % slope == key % slope
	for (int32_t key = columns8; (int32_t) order.size() < rows * columns; ++key) {
		const int32_t xr = key % slope;
		// y ranges over cells with x = key - y * slope inside the block
		int32_t ymin = key >= columns ? (key - (columns - 1) + slope - 1) / slope : 0;
		int32_t ymax = key / slope < rows - 1 ? key / slope : rows - 1;
		(void) xr;
		if (key & 1) {
			for (int32_t y = ymin; y <= ymax; ++y) { int32_t x = key - y * slope; if (y >= rows8 || x >= columns8) order.push_back(y << log_columns | x); }
		} else {
			for (int32_t y = ymax; y >= ymin; --y) { int32_t x = key - y * slope; if (y >= rows8 || x >= columns8) order.push_back(y << log_columns | x); }
		}
	}
	out->swap(order);
}

// ------------------------------------------------------------------------------------------------
// forward DCT used to turn the LF samples of a multi-cell varblock into its LLF coefficients
// (Perera-Liu radix-2 DCT-II, unscaled; operation order of j40.h:5764-5800 and 5944-5970)

namespace {

// one column: elements at in[i * stride]; both buffers are clobbered; result in `out`
void fdct_column(float *out, float *in, int32_t t, int32_t stride, const float *hs) {
	const int32_t N = 1 << t;
	if (t == 0) { out[0] = in[0]; return; }
	if (t == 1) { float x = in[0], y = in[stride]; out[0] = x + y; out[stride] = x - y; return; }
	for (int32_t i = 0; i < N / 2; ++i) {
		float x = in[i * stride], y = in[(N - i - 1) * stride];
		out[i * stride] = x + y;
		out[(N / 2 + i) * stride] = (x - y) * hs[N / 2 + i];
	}
	fdct_column(in, out, t - 1, stride, hs);
	fdct_column(in + N / 2 * stride, out + N / 2 * stride, t - 1, stride, hs);
	for (int32_t i = 0; i < N / 2; ++i) out[i * 2 * stride] = in[i * stride];
	out[stride] = 1.4142135623730951f * in[N / 2 * stride] + in[(N / 2 + 1) * stride];
	for (int32_t i = 1; i < N / 2 - 1; ++i) out[(i * 2 + 1) * stride] = in[(N / 2 + i) * stride] + in[(N / 2 + i + 1) * stride];
	out[(N - 1) * stride] = in[(N - 1) * stride];
}

void fdct_columns(float *out, float *in, int32_t t, int32_t rep, const float *hs) {
	for (int32_t r = 0; r < rep; ++r) fdct_column(out + r, in + r, t, rep, hs);
}

} // namespace

void forward_dct2d_scaled_for_llf(float *buf, float *scratch, int32_t log_rows, int32_t log_columns) {
	const float *hs = half_secants(), *sc = lf2llf_scales();
	const int32_t rows = 1 << log_rows, columns = 1 << log_columns;
	fdct_columns(scratch, buf, log_rows, columns, hs);                       // along rows, per column
	for (int32_t y = 0; y < rows; ++y) for (int32_t x = 0; x < columns; ++x) buf[x * rows + y] = scratch[y * columns + x];
	fdct_columns(scratch, buf, log_columns, rows, hs);                       // scratch is [columns][rows]
	for (int32_t y = 0; y < columns; ++y) for (int32_t x = 0; x < rows; ++x) scratch[y * rows + x] *= sc[rows + x] * sc[columns + y];
	if (log_columns > log_rows) {
		for (int32_t y = 0; y < columns; ++y) for (int32_t x = 0; x < rows; ++x) buf[x * columns + y] = scratch[y * rows + x];
	} else {
		memcpy(buf, scratch, sizeof(float) * (size_t) (rows * columns));
	}
}

} // namespace j40hip
