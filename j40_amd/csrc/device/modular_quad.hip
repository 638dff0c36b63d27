// j40_amd/csrc/device/modular_quad.hip -- K3, four sections per wavefront (replaces j40__modular_channel, j40.h:4127-4240, for
// frames with thousands of sections: plan_build.cpp's assign_coop flags them `quad`).
//
// k_modular_coop (modular_coop.hip) gives a wavefront ONE section and runs the section's state on the scalar unit; measured, it
// is bound by instruction issue: a CU has one scalar port and one vector port, and a section's sample costs ~180 instructions on
// them however many wavefronts are resident. Here a wavefront takes FOUR sections, sixteen lanes each, and keeps ALL of a
// section's state in vector registers, replicated over its sixteen lanes -- so every vector instruction advances four streams:
//
//   * tree: lane j of a group holds branch nodes j, j + 16, j + 32, j + 48 and the path masks of the same leaves (DevCoopTree).
//     The fifteen property values are spread over the group's lanes (lane k holds property k), each lane fetches the property
//     its node tests with ds_bpermute, compares, and a ballot gives every branch outcome of all four trees at once; each group
//     shifts its sixteen bits out of it. A second ballot finds the leaf whose path matches (exactly one per group).
//   * the leaf's record (predictor | hybrid config | max token, alias table, offset, multiplier) is one 16-byte LDS read;
//   * rANS state, bit accumulator (64 bits, refilled from a word requested one refill ahead), hybrid integer, prediction: plain
//     per-lane code, identical in the sixteen lanes of a group, divergent between groups only where their streams differ
//     (refill, extra bits, predictor);
//   * alias tables of the code spec in LDS, one copy per workgroup of four wavefronts (sixteen sections); the three most recent
//     rows of the channel being decoded in LDS per group; decoded samples leave sixteen at a time (32-byte stores).
//
// Groups step in lockstep, one sample each per iteration, like k_hf_lanes' lanes. Integer work: bit-exact with the reference.
#include <hip/hip_runtime.h>
#include "modular_dev.h"
#include "kernels.h"

namespace j40hip {

enum { QUAD_WAVES = 4, QUAD_LEAF_BYTES = 64 * 16 };

struct QuadBits {
	const uint32_t *words; uint32_t last_word, kw, ahead;
	uint64_t acc; int32_t nb;
	uint32_t remaining, consumed, err;
};
J40_DEV uint32_t quad_word(const QuadBits &b, uint32_t i) { return b.words[i < b.last_word ? i : b.last_word]; }
J40_DEV void quad_refill(QuadBits &b) {   // keeps >= 32 bits in the accumulator
	if (b.nb < 32) { b.acc |= (uint64_t) b.ahead << b.nb; b.nb += 32; ++b.kw; b.ahead = quad_word(b, b.kw); }
}
J40_DEV uint32_t quad_take(QuadBits &b, int32_t n) {   // n in [0, 32), nb >= 32 on entry
	if ((uint32_t) n > b.remaining) { if (!b.err) b.err = ERR_SHRT; b.remaining = 0; return 0; }
	const uint32_t v = (uint32_t) b.acc & ((1u << n) - 1);
	b.acc >>= n; b.nb -= n; b.remaining -= (uint32_t) n; b.consumed += (uint32_t) n;
	return v;
}

J40_DEV int32_t quad_predict(int32_t predictor, int32_t w, int32_t n, int32_t nw, int32_t ne, int32_t nn, int32_t nee, int32_t ww) {  // j40.h:4080
	switch (predictor) {
	case 0: return 0;
	case 1: return w;
	case 2: return n;
	case 3: return (w + n) / 2;
	case 4: return mod_abs(n - nw) < mod_abs(w - nw) ? w : n;
	case 5: return mod_gradient(w, n, nw);
	case 7: return ne;
	case 8: return nw;
	case 9: return ww;
	case 10: return (w + nw) / 2;
	case 11: return (n + nw) / 2;
	case 12: return (n + ne) / 2;
	default: return (6 * n - 2 * nn + 7 * w + ww + nee + 3 * ne + 8) / 16;   // 13 (the host admits no other)
	}
}

__global__ void __launch_bounds__(64 * QUAD_WAVES) k_modular_quad(DevModPlan plan, int32_t first_section, int32_t num_sections, int32_t spec_idx, int32_t rows_width) {
	extern __shared__ __attribute__((aligned(16))) uint8_t quad_lds[];
	const int32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, j = lane & 15;
	const DevCodeSpec &spec = plan.spec[spec_idx];
	const uint32_t span = spec.table_span, base_off = plan.clusters[spec.cluster_off].table_off;
	const int32_t log_bucket = 12 - spec.log_alpha_size;
	uint64_t *l_alias = (uint64_t *) quad_lds;
	for (uint32_t i = (uint32_t) tid; i < span; i += 64 * QUAD_WAVES) l_alias[i] = plan.pool_u64[base_off + i];
	const uint32_t group_bytes = QUAD_LEAF_BYTES + 12u * (uint32_t) rows_width;
	uint8_t *area = quad_lds + ((span * 8u + 15u) & ~15u) + (uint32_t) (wave * 4 + g) * group_bytes;
	uint4 *l_leaf = (uint4 *) area;
	int32_t *l_rows = (int32_t *) (area + QUAD_LEAF_BYTES);

	const int32_t s = first_section + ((int32_t) blockIdx.x * QUAD_WAVES + wave) * 4 + g;
	bool active = s < first_section + num_sections;
	DevModSection sec;
	sec.quad = 0; sec.coop_idx = -1; sec.num_channels = 0; sec.byte_off = sec.size = sec.bit_off = 0; sec.sidx = 0; sec.sub_off = sec.chan_off = -1; sec.first_channel = 0;
	sec.gx = sec.gy = sec.gw = sec.gh = 0;
	if (active) sec = plan.sections[s];
	active = active && sec.quad != 0 && sec.coop_idx >= 0;
	const bool mine = active;   // this group reports the section's status
	// this lane's four branch nodes and the path masks of its four leaves; the leaf records into LDS
	int32_t np[4] = {-1, -1, -1, -1}, nt[4] = {0, 0, 0, 0};
	uint32_t mlo[4] = {0, 0, 0, 0}, mhi[4] = {0, 0, 0, 0}, wlo[4] = {1, 1, 1, 1}, whi[4] = {0, 0, 0, 0};
	uint32_t used = 0; int32_t slots = 0;
	if (active) {
		const DevCoopTree *tree = plan.coop_trees + sec.coop_idx;
		used = tree->used_props;
		const int32_t most = tree->num_nodes > tree->num_leaves ? tree->num_nodes : tree->num_leaves;
		slots = (most + 15) >> 4;
		for (int k = 0; k < 4; ++k) {
			const int32_t i = 16 * k + j;
			np[k] = tree->node_prop[i]; nt[k] = tree->node_thr[i];
			mlo[k] = tree->mask_lo[i]; mhi[k] = tree->mask_hi[i]; wlo[k] = tree->want_lo[i]; whi[k] = tree->want_hi[i];
			l_leaf[i] = make_uint4(tree->leaf_a[i], tree->leaf_tab[i] - base_off, (uint32_t) tree->leaf_off[i], (uint32_t) tree->leaf_mul[i]);
		}
	}
	// wave-uniform: the most slots any group needs, the properties any group tests
	int32_t nslots = 0; uint32_t used_any = 0;
	for (int q = 0; q < 4; ++q) {
		const int32_t sl = __builtin_amdgcn_readlane(slots, 16 * q); nslots = sl > nslots ? sl : nslots;
		used_any |= (uint32_t) __builtin_amdgcn_readlane((int32_t) used, 16 * q);
	}
	const DevModFrame *fp = plan.frame;
	const int32_t check_end = fp->check_section_end; const uint32_t declared_end = fp->single_declared_end;

	QuadBits b;
	b.words = (const uint32_t *) plan.codestream; b.err = 0;
	{
		const uint32_t p0 = sec.byte_off * 8 + sec.bit_off;
		b.last_word = (sec.byte_off + sec.size + 3) >> 2;
		b.kw = p0 >> 5;
		b.acc = 0; b.nb = 0; b.ahead = 0;
		if (active) { b.acc = (uint64_t) (quad_word(b, b.kw) >> (p0 & 31)); b.nb = 32 - (int32_t) (p0 & 31); ++b.kw; b.ahead = quad_word(b, b.kw); }
		b.remaining = sec.bit_off <= sec.size * 8 ? sec.size * 8 - sec.bit_off : 0;
		b.consumed = p0;
	}
	uint32_t state = 0, err = 0;
	// channel being decoded
	int32_t cidx = -1, gw = 0, gh = 0, stride = 0, x = 0, y = 0;
	int16_t *base = nullptr;
	int32_t o_cur = 0, o_prev = 2 * rows_width, o_pprev = rows_width;
	int32_t r_nww = 0, r_nw = 0, r_n = 0, r_ne = 0, r_nee = 0, c_w = 0, c_ww = 0, vout = 0;
	bool decoding = false;   // a sample of channel cidx is next
	auto next_channel = [&]() {   // the next channel with samples, or the end of the section
		decoding = false;
		while (++cidx < sec.num_channels) {
			const ModChan chan = mod_channel(plan, sec, cidx);
			if (chan.gw <= 0 || chan.gh <= 0) continue;
			base = chan.base; stride = chan.stride; gw = chan.gw; gh = chan.gh;
			x = 0; y = 0; o_cur = 0; o_prev = 2 * rows_width; o_pprev = rows_width;
			r_nww = r_nw = r_n = r_ne = r_nee = c_w = c_ww = 0;
			decoding = true;
			break;
		}
	};
	if (active) { next_channel(); active = decoding; }
	__syncthreads();   // alias tables staged

	while (__builtin_amdgcn_ballot_w64(active)) {
		if (active) {
			const int32_t *prev = l_rows + o_prev, *pprev = l_rows + o_pprev;
			const int32_t vnn = pprev[x], r_next = prev[x + 3];
			const int32_t pw = x > 0 ? c_w : y > 0 ? r_n : 0;
			const int32_t pn = y > 0 ? r_n : pw;
			const int32_t pnw = x > 0 && y > 0 ? r_nw : pw;
			const int32_t pne = x + 1 < gw && y > 0 ? r_ne : pn;
			const int32_t pnn = y > 1 ? vnn : pn;
			const int32_t pnee = x + 2 < gw && y > 0 ? r_nee : pne;
			const int32_t pww = x > 1 ? c_ww : pw;
			const int32_t pnww = x > 1 && y > 0 ? r_nww : pww;
			// lane k of the group holds the value of property k (j40.h:4141-4155)
			int32_t propreg = 0;
			if (used_any & (1u << 0)) propreg = j == 0 ? cidx : propreg;
			if (used_any & (1u << 1)) propreg = j == 1 ? sec.sidx : propreg;
			if (used_any & (1u << 2)) propreg = j == 2 ? y : propreg;
			if (used_any & (1u << 3)) propreg = j == 3 ? x : propreg;
			if (used_any & (1u << 4)) propreg = j == 4 ? mod_abs(pn) : propreg;
			if (used_any & (1u << 5)) propreg = j == 5 ? mod_abs(pw) : propreg;
			if (used_any & (1u << 6)) propreg = j == 6 ? pn : propreg;
			if (used_any & (1u << 7)) propreg = j == 7 ? pw : propreg;
			if (used_any & (1u << 8)) propreg = j == 8 ? (x > 0 ? pw - (pww + pnw - pnww) : pw) : propreg;
			if (used_any & (1u << 9)) propreg = j == 9 ? pw + pn - pnw : propreg;
			if (used_any & (1u << 10)) propreg = j == 10 ? pw - pnw : propreg;
			if (used_any & (1u << 11)) propreg = j == 11 ? pnw - pn : propreg;
			if (used_any & (1u << 12)) propreg = j == 12 ? pn - pne : propreg;
			if (used_any & (1u << 13)) propreg = j == 13 ? pn - pnn : propreg;
			if (used_any & (1u << 14)) propreg = j == 14 ? pw - pww : propreg;
			// every branch's outcome, sixteen per slot and group
			uint32_t out_lo = 0, out_hi = 0;
			for (int k = 0; k < 4; ++k) if (k < nslots) {
				const int32_t val = __builtin_amdgcn_ds_bpermute((16 * g + (np[k] & 15)) << 2, propreg);
				const uint64_t m = __builtin_amdgcn_ballot_w64(np[k] >= 0 && val > nt[k]);
				const uint32_t bits = (uint32_t) (m >> (16 * g)) & 0xffffu;
				if (k == 0) out_lo |= bits; else if (k == 1) out_lo |= bits << 16; else if (k == 2) out_hi |= bits; else out_hi |= bits << 16;
			}
			int32_t leaf = 0;
			for (int k = 0; k < 4; ++k) if (k < nslots) {
				const uint64_t m = __builtin_amdgcn_ballot_w64(((out_lo & mlo[k]) == wlo[k]) & ((out_hi & mhi[k]) == whi[k]));
				const uint32_t bits = (uint32_t) (m >> (16 * g)) & 0xffffu;
				if (bits) leaf = 16 * k + (int32_t) __builtin_ctz(bits);
			}
			const uint4 rec = l_leaf[leaf];
			const uint32_t la = rec.x;
			// one rANS symbol (j40.h:2441)
			quad_refill(b);
			if (state == 0) { state = quad_take(b, 16); state |= quad_take(b, 16) << 16; quad_refill(b); }
			const uint32_t idx = state & 0xfff, bucket = idx >> log_bucket, pos = idx & ((1u << log_bucket) - 1);
			const uint64_t e = l_alias[rec.y + bucket];
			const bool aliased = pos >= (uint32_t) (e & 0xff);
			int32_t token = (int32_t) (aliased ? (uint32_t) (e >> 20) & 0xff : bucket);
			const uint32_t offset = aliased ? (uint32_t) (e >> 8) & 0xfff : 0;
			const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (uint32_t) (e >> 41) & 0x1fff;
			state = d * (state >> 12) + offset + pos;
			if (state < (1u << 16)) state = (state << 16) | quad_take(b, 16);
			// hybrid integer (j40.h:2313)
			const int32_t split_exp = (int32_t) ((la >> 4) & 15), msb = (int32_t) ((la >> 8) & 15), lsb = (int32_t) ((la >> 12) & 15), max_token = (int32_t) (la >> 16);
			const int32_t split = 1 << split_exp;
			int32_t v = token;
			if (token >= split) {
				if (token > max_token) { token = max_token; if (!b.err) b.err = ERR_IOVF; }
				const int32_t in_token = msb + lsb;
				const int32_t midbits = split_exp - in_token + ((token - split) >> in_token);
				quad_refill(b);
				const int32_t mid = (int32_t) quad_take(b, midbits & 31);
				const int32_t top = 1 << msb;
				const int32_t lo = token & ((1 << lsb) - 1), hi = (token >> lsb) & (top - 1);
				v = ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
			}
			v = ((v & 1) ? -(v / 2 + 1) : v / 2) * (int32_t) rec.w + (int32_t) rec.z;
			v += quad_predict((int32_t) (la & 15), pw, pn, pnw, pne, pnn, pnee, pww);
			if (v < -32768 || v > 32767) err = ERR_POVF;
			if (!err) {
				l_rows[o_cur + x] = v;
				vout = j == (x & 15) ? v : vout;
				if (((x & 15) == 15 || x + 1 == gw) && j <= (x & 15)) (base + (size_t) y * (size_t) stride)[(x & ~15) + j] = (int16_t) vout;
				c_ww = c_w; c_w = v; r_nww = r_nw; r_nw = r_n; r_n = r_ne; r_ne = r_nee; r_nee = r_next;
				if (++x == gw) {
					x = 0;
					if (++y == gh) next_channel();
					else {
						const int32_t t = o_cur; o_cur = o_pprev; o_pprev = o_prev; o_prev = t;
						r_n = l_rows[o_prev]; r_ne = l_rows[o_prev + 1]; r_nee = l_rows[o_prev + 2];
						r_nww = r_nw = c_w = c_ww = 0;
					}
				}
			}
			active = decoding && !err && !b.err;
		}
	}
	if (mine) {
		uint32_t status = b.err ? b.err : err;
		if (!status) {   // the stream's final state (j40.h:2884)
			quad_refill(b);
			if (state) { if (state != 0x130000) status = ERR_ANS; }
			else { if (quad_take(b, 16) != 0x0000) status = ERR_ANS; if (quad_take(b, 16) != 0x0013) status = ERR_ANS; if (b.err) status = b.err; }
		}
		if (!status && check_end) {   // frames that are a single section end exactly here (entropy_dev.h, bits_finish_section)
			quad_refill(b);
			const int32_t pad = (int32_t) ((0u - b.consumed) & 7);
			if (quad_take(b, pad)) status = ERR_PAD0;
			if (b.err) status = b.err;
			const uint32_t at = b.consumed >> 3;
			if (!status) { if (at < declared_end) status = ERR_SHRT; else if (at > declared_end) status = ERR_EXCS; }
		}
		if (j == 0) plan.status[s] = status;
	}
}

void launch_modular_quad(const DevModPlan &plan, int32_t first_section, int32_t num_sections, int32_t spec_idx, uint32_t table_span, int32_t max_width, hipStream_t stream) {
	if (num_sections <= 0) return;
	const int32_t rows_width = max_width + 8;
	const size_t lds = ((size_t) table_span * 8 + 15) / 16 * 16 + (size_t) QUAD_WAVES * 4 * (QUAD_LEAF_BYTES + 12 * (size_t) rows_width);
	static bool configured = false;
	if (!configured) { (void) hipFuncSetAttribute((const void *) k_modular_quad, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
	const unsigned blocks = (unsigned) ((num_sections + 4 * QUAD_WAVES - 1) / (4 * QUAD_WAVES));
	hipLaunchKernelGGL(k_modular_quad, dim3(blocks), dim3(64 * QUAD_WAVES), lds, stream, plan, first_section, num_sections, spec_idx, rows_width);
}

} // namespace j40hip
