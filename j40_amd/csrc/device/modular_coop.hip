// j40_amd/csrc/device/modular_coop.hip -- K3, wave-cooperative form: ONE Modular section per wavefront, decoded by all 64 lanes
// together (replaces j40__modular_channel, j40.h:4127-4240, for the sections plan_build.cpp's assign_coop selects).
//
// The per-sample loop of a Modular stream is serial (the MA tree is walked with properties of the samples just decoded, and the
// rANS state chains every symbol to the one before), so a wavefront cannot decode 64 samples of one stream at once. What it can
// do is shorten the serial chain of ONE sample, which k_modular_sections spends mostly waiting on dependent LDS reads -- four
// words per tree level, the cluster map, the cluster record, the alias entry, two row reads:
//
//   * the tree is not walked: lane i holds branch node i and evaluates its test for the current sample; one ballot is the
//     outcome of every branch at once, and leaf i is the one reached iff the outcomes of its ancestors are those on its path,
//     (outcomes & mask_i) == want_i -- a second ballot, whose lowest set bit names the leaf (DevCoopTree, plan.h);
//   * the leaf's predictor, offset, multiplier and its cluster's hybrid-integer configuration and alias table sit in lane
//     `leaf` of five registers: five v_readlane, no memory;
//   * the codestream is read 64 words at a time, one word per lane (a coalesced 256-byte request issued 64 words ahead of its
//     use); the decoder's next word is a v_readlane;
//   * the row above and the row above that are held 64 samples per register (lane = column mod 64), refilled from the LDS row
//     ring once per 64 samples; the decoded samples collect in a register (v_writelane) and leave as one coalesced store per 64.
//
// Per sample that leaves ONE memory access on the serial chain: the alias entry, a scalar load (wave-uniform address, 8 bytes).
// Everything else is scalar-unit arithmetic on wave-uniform values. Integer work: bit-exact with the reference.
#include <hip/hip_runtime.h>
#include "modular_coop_dev.h"
#include "kernels.h"

namespace j40hip {

// MODE 0: neighbours, properties and prediction on the vector ALU; MODE 1: neighbours and prediction on the scalar unit, only the
// fifteen property values (which feed per-lane selects anyway) on the vector ALU
template <int MODE>
__global__ void __launch_bounds__(64) k_modular_coop(DevModPlan plan, int32_t first_section, int32_t rows_width) {
	extern __shared__ int32_t coop_rows[];   // [3][rows_width]: the three most recent rows of the channel being decoded
	const uint32_t lane = threadIdx.x;
	const int32_t s = first_section + (int32_t) blockIdx.x;
	const DevModSection sec = plan.sections[s];
	const int32_t coop = coop_sc(sec.coop_idx);
	if (coop < 0 || coop_sc(sec.quad)) return;   // k_modular_sections' section (it also reports the sections that failed on the host), or k_modular_quad's
	const DevModFrame *fp = plan.frame;
	const int32_t check_end = coop_sc(fp->check_section_end); const uint32_t declared_end = (uint32_t) coop_sc((int32_t) fp->single_declared_end);
	const CoopTreeRegs t = coop_load_tree(plan.coop_trees + coop, lane);
	const int32_t log_bucket = 12 - coop_sc(plan.spec[coop_sc(sec.spec_idx)].log_alpha_size);
	const CoopConstU64 alias = (CoopConstU64) coop_sc_ptr(plan.pool_u64);
	const int32_t sidx = coop_sc(sec.sidx), num_channels = coop_sc(sec.num_channels);

	CoopBits b;
	coop_bits_init(b, coop_sc_ptr(plan.codestream), (uint32_t) coop_sc((int32_t) sec.byte_off), (uint32_t) coop_sc((int32_t) sec.size), (uint32_t) coop_sc((int32_t) sec.bit_off), lane);
	uint32_t state = 0, err = 0;

	for (int32_t cidx = 0; cidx < num_channels && !b.err && !err; ++cidx) {
		const ModChan chan = mod_channel(plan, sec, cidx);
		const int32_t stride = coop_sc(chan.stride), gw = coop_sc(chan.gw), gh = coop_sc(chan.gh);
		if (gw <= 0 || gh <= 0) continue;
		coop_decode_channel<MODE>(b, state, err, t, alias, log_bucket, cidx, sidx, coop_sc_ptr(chan.base), stride, gw, gh, coop_rows, rows_width, lane);
	}
	uint32_t status = b.err ? b.err : err;
	if (!status) status = coop_finish_code(b, state, lane);
	if (!status && check_end) {   // frames that are a single section end exactly here (entropy_dev.h, bits_finish_section)
		const int32_t pad = (int32_t) ((0u - b.consumed) & 7);
		if (coop_take(b, pad, lane)) status = ERR_PAD0;
		if (b.err) status = b.err;
		const uint32_t at = b.consumed >> 3;
		if (!status) { if (at < declared_end) status = ERR_SHRT; else if (at > declared_end) status = ERR_EXCS; }
	}
	if (lane == 0) plan.status[s] = status;
}

void launch_modular_coop(const DevModPlan &plan, int32_t first_section, int32_t num_sections, int32_t max_width, hipStream_t stream) {
	if (num_sections <= 0) return;
	const int32_t rows_width = ((max_width + 63) & ~63) + 64;
	static const int mode = [] { const char *e = getenv("J40HIP_COOP_MODE"); return e ? atoi(e) : 0; }();
	if (mode == 1) hipLaunchKernelGGL(k_modular_coop<1>, dim3((unsigned) num_sections), dim3(64), (size_t) rows_width * 12, stream, plan, first_section, rows_width);
	else hipLaunchKernelGGL(k_modular_coop<0>, dim3((unsigned) num_sections), dim3(64), (size_t) rows_width * 12, stream, plan, first_section, rows_width);
}

} // namespace j40hip
