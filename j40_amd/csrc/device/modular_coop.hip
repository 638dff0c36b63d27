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
#include "modular_dev.h"
#include "kernels.h"

namespace j40hip {

typedef const __attribute__((address_space(4))) uint64_t *CoopConstU64;   // read-only for the kernel's lifetime: scalar loads

J40_DEV int32_t coop_rl(int32_t v, int32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
J40_DEV int32_t coop_sc(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> J40_DEV T *coop_sc_ptr(T *p) {
	const uint64_t v = (uint64_t) p;
	return (T *) ((uint64_t) (uint32_t) coop_sc((int32_t) (uint32_t) v) | (uint64_t) (uint32_t) coop_sc((int32_t) (uint32_t) (v >> 32)) << 32);
}

// the section's bits; every member but cur / nxt is wave-uniform. Same rules as DevBits (entropy_dev.h): bits past the section's
// last byte are never handed out, asking for them is `shrt`.
struct CoopBits {
	const uint32_t *words;      // the codestream as 32-bit words (4-byte aligned, padded)
	uint32_t last_word;         // highest index that may be requested
	uint32_t wbase;             // word index held by lane 0 of `cur`
	uint32_t kw;                // next word to enter the accumulator
	uint32_t cur, nxt;          // per lane: words wbase + lane, wbase + 64 + lane
	uint64_t acc; int32_t nb;   // accumulator: nb valid bits
	uint32_t remaining;         // bits of the section not yet handed out
	uint32_t consumed;          // absolute bit position
	uint32_t err;
};

J40_DEV uint32_t coop_fetch(const CoopBits &b, uint32_t first, uint32_t lane) {
	const uint32_t i = first + lane;
	return b.words[i < b.last_word ? i : b.last_word];
}
J40_DEV void coop_bits_init(CoopBits &b, const uint8_t *codestream, uint32_t byte_off, uint32_t size, uint32_t bit_off, uint32_t lane) {
	b.words = (const uint32_t *) codestream;
	const uint32_t p0 = byte_off * 8 + bit_off;
	b.last_word = (byte_off + size + 3) >> 2;
	b.wbase = b.kw = p0 >> 5;
	b.cur = coop_fetch(b, b.wbase, lane); b.nxt = coop_fetch(b, b.wbase + 64, lane);
	b.acc = (uint64_t) ((uint32_t) coop_rl((int32_t) b.cur, 0) >> (p0 & 31)); b.nb = 32 - (int32_t) (p0 & 31); ++b.kw;
	b.remaining = bit_off <= size * 8 ? size * 8 - bit_off : 0;
	b.consumed = p0; b.err = 0;
}
J40_DEV void coop_refill(CoopBits &b, uint32_t lane) {   // nb <= 32 on entry
	uint32_t i = b.kw - b.wbase;
	if (i >= 64) { b.cur = b.nxt; b.wbase += 64; b.nxt = coop_fetch(b, b.wbase + 64, lane); i -= 64; }
	const uint32_t w = (uint32_t) coop_rl((int32_t) b.cur, (int32_t) i);
	++b.kw;
	b.acc |= (uint64_t) w << b.nb; b.nb += 32;
}
J40_DEV uint32_t coop_take(CoopBits &b, int32_t n, uint32_t lane) {   // n in [0, 32)
	if (b.nb < n) coop_refill(b, lane);
	if ((uint32_t) n > b.remaining) { if (!b.err) b.err = ERR_SHRT; b.remaining = 0; return 0; }
	const uint32_t v = (uint32_t) b.acc & ((1u << n) - 1);
	b.acc >>= n; b.nb -= n; b.remaining -= (uint32_t) n; b.consumed += (uint32_t) n;
	return v;
}

J40_DEV int32_t coop_predict(int32_t predictor, int32_t w, int32_t n, int32_t nw, int32_t ne, int32_t nn, int32_t nee, int32_t ww) {  // j40.h:4080
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
	const DevCoopTree *tree = plan.coop_trees + coop;
	const uint32_t used = (uint32_t) coop_sc((int32_t) tree->used_props);
	// this lane's branch node and leaf
	const int32_t my_prop = tree->node_prop[lane], my_thr = tree->node_thr[lane];
	const uint32_t my_mlo = tree->mask_lo[lane], my_mhi = tree->mask_hi[lane], my_wlo = tree->want_lo[lane], my_whi = tree->want_hi[lane];
	const int32_t leaf_a = (int32_t) tree->leaf_a[lane], leaf_tab = (int32_t) tree->leaf_tab[lane], leaf_off = tree->leaf_off[lane], leaf_mul = tree->leaf_mul[lane];
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
		int16_t *base = coop_sc_ptr(chan.base);
		for (int32_t y = 0; y < gh && !b.err && !err; ++y) {
			int16_t *row = base + (size_t) y * (size_t) stride;
			int32_t *cur = coop_rows + (y % 3) * rows_width;
			const int32_t *prev = coop_rows + ((y + 2) % 3) * rows_width, *pprev = coop_rows + ((y + 1) % 3) * rows_width;
			// the row above, 64 samples per register: vprev = columns xb .. xb + 63, vprev2 = the 64 behind them (NEE and the
			// sample that becomes NEE reach up to three columns ahead); vpp = the row above that. Rows that do not exist
			// yet (y < 2) and columns past the rectangle hold stale values that are never selected.
			int32_t vprev = prev[lane], vprev2 = 0, vpp = 0, vout = 0;
			int32_t r_nww = 0, r_nw = 0, r_n = coop_rl(vprev, 0), r_ne = coop_rl(vprev, 1), r_nee = coop_rl(vprev, 2), c_w = 0, c_ww = 0;
			for (int32_t xb = 0; xb < gw && !b.err && !err; xb += 64) {
				if (xb) vprev = vprev2;
				vprev2 = prev[xb + 64 + (int32_t) lane];
				vpp = pprev[xb + (int32_t) lane];
				const int32_t nblk = gw - xb < 64 ? gw - xb : 64;
				for (int32_t i = 0; i < nblk; ++i) {
					// The scalar unit is the port this kernel saturates (one scalar instruction per cycle per CU, shared by its four
					// SIMDs, against one vector instruction per cycle per CU): the sample side of the loop -- neighbours, properties,
					// prediction -- is therefore kept in vector registers (every lane computes the same value), which the compiler
					// does as soon as the column index and the values read from the rows are not known to be uniform. The
					// entropy side (rANS state, bit accumulator, hybrid integer) stays on the scalar unit.
					int32_t x = xb + i, vnn = coop_rl(vpp, i), r_next = i + 3 < 64 ? coop_rl(vprev, i + 3) : coop_rl(vprev2, i + 3 - 64);
					if (MODE == 0) { asm("" : "+v"(x)); asm("" : "+v"(vnn)); asm("" : "+v"(r_next)); }
					const int32_t pw = x > 0 ? c_w : y > 0 ? r_n : 0;
					const int32_t pn = y > 0 ? r_n : pw;
					const int32_t pnw = x > 0 && y > 0 ? r_nw : pw;
					const int32_t pne = x + 1 < gw && y > 0 ? r_ne : pn;
					const int32_t pnn = y > 1 ? vnn : pn;
					const int32_t pnee = x + 2 < gw && y > 0 ? r_nee : pne;
					const int32_t pww = x > 1 ? c_ww : pw;
					const int32_t pnww = x > 1 && y > 0 ? r_nww : pww;
					// every branch's outcome: the value of the property this lane's node tests (j40.h:4141-4155), then one compare
					int32_t qx = x, qw = pw, qn = pn, qnw = pnw, qne = pne, qnn = pnn, qww = pww, qnww = pnww;
					if (MODE == 1) { asm("" : "+v"(qx)); asm("" : "+v"(qw)); asm("" : "+v"(qn)); asm("" : "+v"(qnw)); asm("" : "+v"(qne)); asm("" : "+v"(qnn)); asm("" : "+v"(qww)); asm("" : "+v"(qnww)); }
					int32_t myval = 0;
					if (used & (1u << 0)) myval = my_prop == 0 ? cidx : myval;
					if (used & (1u << 1)) myval = my_prop == 1 ? sidx : myval;
					if (used & (1u << 2)) myval = my_prop == 2 ? y : myval;
					if (used & (1u << 3)) myval = my_prop == 3 ? qx : myval;
					if (used & (1u << 4)) myval = my_prop == 4 ? mod_abs(qn) : myval;
					if (used & (1u << 5)) myval = my_prop == 5 ? mod_abs(qw) : myval;
					if (used & (1u << 6)) myval = my_prop == 6 ? qn : myval;
					if (used & (1u << 7)) myval = my_prop == 7 ? qw : myval;
					if (used & (1u << 8)) myval = my_prop == 8 ? (qx > 0 ? qw - (qww + qnw - qnww) : qw) : myval;
					if (used & (1u << 9)) myval = my_prop == 9 ? qw + qn - qnw : myval;
					if (used & (1u << 10)) myval = my_prop == 10 ? qw - qnw : myval;
					if (used & (1u << 11)) myval = my_prop == 11 ? qnw - qn : myval;
					if (used & (1u << 12)) myval = my_prop == 12 ? qn - qne : myval;
					if (used & (1u << 13)) myval = my_prop == 13 ? qn - qnn : myval;
					if (used & (1u << 14)) myval = my_prop == 14 ? qw - qww : myval;
					const uint64_t outcomes = __builtin_amdgcn_ballot_w64(my_prop >= 0 && myval > my_thr);
					const uint64_t reached = __builtin_amdgcn_ballot_w64((((uint32_t) outcomes & my_mlo) == my_wlo) & (((uint32_t) (outcomes >> 32) & my_mhi) == my_whi));
					const int32_t leaf = (int32_t) __builtin_ctzll(reached);   // exactly one leaf matches
					const uint32_t la = (uint32_t) coop_rl(leaf_a, leaf), tab = (uint32_t) coop_rl(leaf_tab, leaf);
					const int32_t off = coop_rl(leaf_off, leaf), mul = coop_rl(leaf_mul, leaf);
					// one rANS symbol (j40.h:2441)
					if (state == 0) { state = coop_take(b, 16, lane); state |= coop_take(b, 16, lane) << 16; }
					const uint32_t idx = state & 0xfff, bucket = idx >> log_bucket, pos = idx & ((1u << log_bucket) - 1);
					const uint64_t e = alias[tab + bucket];
					const bool aliased = pos >= (uint32_t) (e & 0xff);
					int32_t token = (int32_t) (aliased ? (uint32_t) (e >> 20) & 0xff : bucket);
					const uint32_t offset = aliased ? (uint32_t) (e >> 8) & 0xfff : 0;
					const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (uint32_t) (e >> 41) & 0x1fff;
					state = d * (state >> 12) + offset + pos;
					if (state < (1u << 16)) state = (state << 16) | coop_take(b, 16, lane);
					// hybrid integer (j40.h:2313)
					const int32_t split_exp = (int32_t) ((la >> 4) & 15), msb = (int32_t) ((la >> 8) & 15), lsb = (int32_t) ((la >> 12) & 15), max_token = (int32_t) (la >> 16);
					const int32_t split = 1 << split_exp;
					int32_t v = token;
					if (token >= split) {
						if (token > max_token) { token = max_token; if (!b.err) b.err = ERR_IOVF; }
						const int32_t in_token = msb + lsb;
						const int32_t midbits = split_exp - in_token + ((token - split) >> in_token);
						const int32_t mid = (int32_t) coop_take(b, midbits & 31, lane);
						const int32_t top = 1 << msb;
						const int32_t lo = token & ((1 << lsb) - 1), hi = (token >> lsb) & (top - 1);
						v = ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
					}
					v = ((v & 1) ? -(v / 2 + 1) : v / 2) * mul + off;
					v += coop_predict((int32_t) (la & 15), pw, pn, pnw, pne, pnn, pnee, pww);
					if (v < -32768 || v > 32767) { err = ERR_POVF; break; }
					vout = (int32_t) lane == i ? v : vout;
					c_ww = c_w; c_w = v; r_nww = r_nw; r_nw = r_n; r_n = r_ne; r_ne = r_nee; r_nee = r_next;
					if (b.err) break;
				}
				if ((int32_t) lane < nblk) { cur[xb + (int32_t) lane] = vout; row[xb + (int32_t) lane] = (int16_t) vout; }
			}
		}
	}
	uint32_t status = b.err ? b.err : err;
	if (!status) {   // the stream's final state (j40.h:2884)
		if (state) { if (state != 0x130000) status = ERR_ANS; }
		else { if (coop_take(b, 16, lane) != 0x0000) status = ERR_ANS; if (coop_take(b, 16, lane) != 0x0013) status = ERR_ANS; if (b.err) status = b.err; }
	}
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
