// j40_amd/csrc/device/modular_coop_dev.h -- the wave-cooperative Modular decoder as device functions (see modular_coop.hip for the
// design): bit reader over a register-resident window of the codestream, and the decode of ONE channel by all 64 lanes of a
// wavefront. Shared by k_modular_coop (sections of Modular frames) and k_lf_groups (the two Modular sub-images of an LfGroup
// section of a VarDCT frame, lf_decode.hip).
#pragma once
#include "modular_dev.h"

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

// this lane's branch node and leaf of a DevCoopTree
struct CoopTreeRegs {
	int32_t my_prop, my_thr;
	uint32_t my_mlo, my_mhi, my_wlo, my_whi;
	int32_t leaf_a, leaf_tab, leaf_off, leaf_mul;
	uint32_t used;   // wave-uniform: the properties the tree tests
};
J40_DEV CoopTreeRegs coop_load_tree(const DevCoopTree *tree, uint32_t lane) {
	CoopTreeRegs t;
	t.used = (uint32_t) coop_sc((int32_t) tree->used_props);
	t.my_prop = tree->node_prop[lane]; t.my_thr = tree->node_thr[lane];
	t.my_mlo = tree->mask_lo[lane]; t.my_mhi = tree->mask_hi[lane]; t.my_wlo = tree->want_lo[lane]; t.my_whi = tree->want_hi[lane];
	t.leaf_a = (int32_t) tree->leaf_a[lane]; t.leaf_tab = (int32_t) tree->leaf_tab[lane]; t.leaf_off = tree->leaf_off[lane]; t.leaf_mul = tree->leaf_mul[lane];
	return t;
}

// One channel (gw x gh samples, rows `stride` apart at `base`) of the stream `b` / `state`, decoded by the whole wavefront.
// coop_rows: [3][rows_width] int32 in LDS, rows_width >= align64(gw) + 64. Every argument but `t`'s per-lane members and `lane`
// is wave-uniform. MODE: see k_modular_coop. PLANE_ROWS: the two rows above are read back from the output plane (coalesced
// 128-byte pieces, once per 64 samples) instead of an LDS ring -- for channels too wide for LDS (the varblock-info channel of an
// LfGroup is two rows of up to 65536 samples); coop_rows is then unused.
template <int MODE, bool PLANE_ROWS = false>
J40_DEV void coop_decode_channel(CoopBits &b, uint32_t &state, uint32_t &err, const CoopTreeRegs &t, CoopConstU64 alias, int32_t log_bucket,
		int32_t cidx, int32_t sidx, int16_t *base, int32_t stride, int32_t gw, int32_t gh, int32_t *coop_rows, int32_t rows_width, uint32_t lane) {
	const uint32_t used = t.used;
	for (int32_t y = 0; y < gh && !b.err && !err; ++y) {
		int16_t *row = base + (size_t) y * (size_t) stride;
		int32_t *cur = coop_rows + (y % 3) * rows_width;
		const int32_t *prev = coop_rows + ((y + 2) % 3) * rows_width, *pprev = coop_rows + ((y + 1) % 3) * rows_width;
		const int16_t *prow = row - stride, *pprow = row - 2 * (size_t) stride;   // PLANE_ROWS (only read where they exist)
		auto above = [&](int32_t x0) { const int32_t k = x0 + (int32_t) lane; return PLANE_ROWS ? (y > 0 && k < gw ? (int32_t) prow[k] : 0) : prev[k]; };
		auto above2 = [&](int32_t x0) { const int32_t k = x0 + (int32_t) lane; return PLANE_ROWS ? (y > 1 && k < gw ? (int32_t) pprow[k] : 0) : pprev[k]; };
		// the row above, 64 samples per register: vprev = columns xb .. xb + 63, vprev2 = the 64 behind them (NEE and the
		// sample that becomes NEE reach up to three columns ahead); vpp = the row above that. Rows that do not exist
		// yet (y < 2) and columns past the rectangle hold stale values that are never selected.
		int32_t vprev = above(0), vprev2 = 0, vpp = 0, vout = 0;
		int32_t r_nww = 0, r_nw = 0, r_n = coop_rl(vprev, 0), r_ne = coop_rl(vprev, 1), r_nee = coop_rl(vprev, 2), c_w = 0, c_ww = 0;
		for (int32_t xb = 0; xb < gw && !b.err && !err; xb += 64) {
			if (xb) vprev = vprev2;
			vprev2 = above(xb + 64);
			vpp = above2(xb);
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
				if (used & (1u << 0)) myval = t.my_prop == 0 ? cidx : myval;
				if (used & (1u << 1)) myval = t.my_prop == 1 ? sidx : myval;
				if (used & (1u << 2)) myval = t.my_prop == 2 ? y : myval;
				if (used & (1u << 3)) myval = t.my_prop == 3 ? qx : myval;
				if (used & (1u << 4)) myval = t.my_prop == 4 ? mod_abs(qn) : myval;
				if (used & (1u << 5)) myval = t.my_prop == 5 ? mod_abs(qw) : myval;
				if (used & (1u << 6)) myval = t.my_prop == 6 ? qn : myval;
				if (used & (1u << 7)) myval = t.my_prop == 7 ? qw : myval;
				if (used & (1u << 8)) myval = t.my_prop == 8 ? (qx > 0 ? qw - (qww + qnw - qnww) : qw) : myval;
				if (used & (1u << 9)) myval = t.my_prop == 9 ? qw + qn - qnw : myval;
				if (used & (1u << 10)) myval = t.my_prop == 10 ? qw - qnw : myval;
				if (used & (1u << 11)) myval = t.my_prop == 11 ? qnw - qn : myval;
				if (used & (1u << 12)) myval = t.my_prop == 12 ? qn - qne : myval;
				if (used & (1u << 13)) myval = t.my_prop == 13 ? qn - qnn : myval;
				if (used & (1u << 14)) myval = t.my_prop == 14 ? qw - qww : myval;
				const uint64_t outcomes = __builtin_amdgcn_ballot_w64(t.my_prop >= 0 && myval > t.my_thr);
				const uint64_t reached = __builtin_amdgcn_ballot_w64((((uint32_t) outcomes & t.my_mlo) == t.my_wlo) & (((uint32_t) (outcomes >> 32) & t.my_mhi) == t.my_whi));
				const int32_t leaf = (int32_t) __builtin_ctzll(reached);   // exactly one leaf matches
				const uint32_t la = (uint32_t) coop_rl(t.leaf_a, leaf), tab = (uint32_t) coop_rl(t.leaf_tab, leaf);
				const int32_t off = coop_rl(t.leaf_off, leaf), mul = coop_rl(t.leaf_mul, leaf);
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
			if ((int32_t) lane < nblk) { if (!PLANE_ROWS) cur[xb + (int32_t) lane] = vout; row[xb + (int32_t) lane] = (int16_t) vout; }
		}
	}
}

// the stream's final state (j40.h:2884); returns the status to report (0: fine)
J40_DEV uint32_t coop_finish_code(CoopBits &b, uint32_t state, uint32_t lane) {
	uint32_t status = 0;
	if (state) { if (state != 0x130000) status = ERR_ANS; }
	else { if (coop_take(b, 16, lane) != 0x0000) status = ERR_ANS; if (coop_take(b, 16, lane) != 0x0013) status = ERR_ANS; if (b.err) status = b.err; }
	return status;
}

} // namespace j40hip
