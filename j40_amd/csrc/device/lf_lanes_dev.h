// j40_amd/csrc/device/lf_lanes_dev.h -- the LfGroup sections of VarDCT frames decoded one section per wavefront LANE
// (k_lf_lanes, lf_decode.hip; SURVEY.md 8f-1: j40__lf_group's two Modular sub-images, j40.h:6722-6790, i.e.
// j40__modular_channel over the LF coefficient image and over the HF metadata image, j40.h:4127-4240).
//
// An LfGroup section is one adaptive stream of a third of a million samples: nothing inside it runs in parallel, and a wavefront
// that spends its 64 lanes on ONE such stream (k_lf_groups, modular_coop_dev.h) ties up 1/64 of the machine per section for a
// quarter of a second -- 3072 sections (256 8K frames) slowed every other kernel they ran beside, and the host's cores were the
// better decoders. Here a lane is a section and a wavefront a frame: the 3072 sections are 256 wavefronts, one per compute unit,
// which the other kernels do not notice. Every iteration each lane decodes one sample of its own stream (same shape as the HF
// coefficient decoder hf_lanes_dev.h, whose bit window and rANS + hybrid-integer step it shares):
//   * the frame's global MA tree and code spec sit in LDS; the walk starts below the nodes that test the channel or stream index
//     (decided once per channel);
//   * W and WW are carried from sample to sample, the row above slides along in registers: one load per sample, issued two samples
//     ahead of its use (NN only for trees and predictors that look at it);
//   * samples are stored as they are decoded, int16, into the frame-wide planes the plan build reads.
// Takes rANS code specs without LZ77 and trees without the weighted predictor or previous-channel properties (what VarDCT encoders
// write for these streams); the host checks that (plan_front.cpp) and decodes the sections itself otherwise.
#pragma once
#include "hf_lanes_dev.h"
#include "modular_dev.h"

namespace j40hip {

// The code tables of a lane's frame. The small ones (context map, hybrid-integer configurations) are staged in LDS like the tree;
// the alias tables (num_clusters << log_alpha entries of 8 bytes: 14 KB for seven clusters of 256 buckets) are read where the
// host put them: staged in LDS they let a wavefront hold three frames (36 of 64 lanes), and two such workgroups on a compute
// unit left no room for the coefficient decoder's (k_hf_lanes, 99 KB), which then ran its workgroups in two rounds.
template <bool ALIAS_LDS> struct LfLaneTablesT {
	const J40_LDS uint8_t *ctx_map;
	const J40_LDS uint32_t *cluster_cfg;
	const J40_GLOBAL uint64_t *alias;
	int32_t log_alpha, log_bucket;
};
template <> struct LfLaneTablesT<true> {
	const J40_LDS uint8_t *ctx_map;
	const J40_LDS uint32_t *cluster_cfg;
	const J40_LDS uint64_t *alias;
	int32_t log_alpha, log_bucket;
};

// what the sections of one frame share (wave-uniform)
struct LfLaneFrame {
	const J40_LDS DevTreeNode *tree;   // the global MA tree
	uint32_t uses;                     // bit 0: some node / leaf looks at NE, 1: NEE, 2: NN, 3: NWW
};

struct LfLane {
	LaneBits b;
	uint32_t state, err, end_bit;
	int32_t chan;                      // 0-2 LF coefficients (Y, X, B), 3 x-from-y, 4 b-from-y, 5 varblock info, 6 sharpness, 7 finished
	int32_t x, y, cw, chh, root;       // position inside the channel, its size, where its tree walk starts
	int32_t pw, pww;                   // the two samples before this one
	int32_t a0, a1, a2, a3, a4;        // the row above at x - 2 .. x + 2 (as stored: fallbacks are applied per sample)
	J40_GLOBAL int16_t *row;           // the channel's current row
	int32_t nb_varblocks;
	bool setup;                        // the next step starts a channel
};

J40_DEV DevTreeNode lf_node(const J40_LDS DevTreeNode *tree, int32_t at) {   // (address-space qualified structs have no copy constructor)
	const J40_LDS int32_t *p = (const J40_LDS int32_t *) (tree + at);
	DevTreeNode n; n.prop = p[0]; n.value = p[1]; n.a = p[2]; n.b = p[3];
	return n;
}
J40_DEV void lf_lane_fail(LfLane &L, uint32_t e) { if (!L.err) L.err = e; L.chan = 7; L.setup = false; }
J40_DEV bool lf_lane_done(const LfLane &L) { return L.chan == 7 && !L.setup; }

J40_DEV uint32_t lf_lane_take(LfLane &L, int32_t n) {   // header bits (n <= 31)
	if (L.b.nbits < n) lane_bits_refill(L.b);
	const uint32_t v = lane_bits_take(L.b, n);
	if (lane_bit_position(L.b) > L.end_bit) lf_lane_fail(L, ERR_SHRT);
	return v;
}

// the stream's final state (j40.h:2884): 0x130000, read now if no symbol was ever decoded
J40_DEV void lf_lane_finish_code(LfLane &L) {
	if (L.state == 0) { lane_bits_refill(L.b); L.state = lf_lane_take(L, 16); L.state |= lf_lane_take(L, 16) << 16; }
	if (!L.err && L.state != 0x130000) lf_lane_fail(L, ERR_ANS);
	L.state = 0;
}

J40_DEV void lf_lane_init(LfLane &L, const J40_GLOBAL DevLfTask &t) {
	lane_bits_init(L.b, (const J40_GLOBAL uint8_t *) t.codestream, 8u * t.byte_off + t.bit_off);
	L.state = 0; L.err = 0; L.end_bit = 8u * (t.byte_off + t.size);
	L.chan = 0; L.setup = true; L.nb_varblocks = 0;
	L.x = L.y = 0; L.cw = L.chh = 0; L.root = 0; L.pw = L.pww = 0; L.a0 = L.a1 = L.a2 = L.a3 = L.a4 = 0; L.row = nullptr;
	if (8u * t.byte_off + t.bit_off > L.end_bit) lf_lane_fail(L, ERR_SHRT);
}

// starts channel L.chan (skipping channels without samples); between the two images: the final state of the first, the varblock
// count and the second image's header (j40.h:6748-6752; only the plain header -- global tree, no transforms: anything else is
// reported as ERR_LFFB and the host decodes the section)
J40_DEV void lf_lane_setup(LfLane &L, const J40_GLOBAL DevLfTask &t, const LfLaneFrame &F) {
	for (;;) {
		if (L.chan == 3) {
			lf_lane_finish_code(L);
			if (L.err) return;
			L.nb_varblocks = (int32_t) lf_lane_take(L, t.nbvb_bits) + 1;
			const uint32_t header = lf_lane_take(L, 4);   // use_global_tree = 1, default wp = 1, no transforms (j40.h:3717-3760)
			if (L.err) return;
			if (header != 3u || 2u * (uint32_t) L.nb_varblocks > t.info_capacity) { lf_lane_fail(L, ERR_LFFB); return; }
		}
		if (L.chan == 7) { lf_lane_finish_code(L); L.chan = 7; L.setup = false; return; }
		int32_t cw, chh; J40_GLOBAL int16_t *base;
		switch (L.chan) {
		case 0: case 1: case 2: cw = t.w8; chh = t.h8; base = (J40_GLOBAL int16_t *) t.lf[L.chan]; break;
		case 3: cw = t.w64; chh = t.h64; base = (J40_GLOBAL int16_t *) t.xfromy; break;
		case 4: cw = t.w64; chh = t.h64; base = (J40_GLOBAL int16_t *) t.bfromy; break;
		case 5: cw = L.nb_varblocks; chh = 2; base = (J40_GLOBAL int16_t *) t.info; break;
		default: cw = t.w8; chh = t.h8; base = (J40_GLOBAL int16_t *) t.sharp; break;
		}
		if (cw <= 0 || chh <= 0) { ++L.chan; continue; }
		L.cw = cw; L.chh = chh; L.row = base; L.x = L.y = 0; L.pw = L.pww = 0; L.a0 = L.a1 = L.a2 = L.a3 = L.a4 = 0;
		// the nodes that test the channel or the stream index lead to the same child for every sample of the channel
		const int32_t cidx = L.chan < 3 ? L.chan : L.chan - 3, sidx = L.chan < 3 ? t.sidx0 : t.sidx2;
		int32_t at = 0;
		for (;;) {
			const DevTreeNode n = lf_node(F.tree, at);
			if (n.prop == 0) at += cidx > n.value ? n.a : n.b;
			else if (n.prop == 1) at += sidx > n.value ? n.a : n.b;
			else break;
		}
		L.root = at;
		L.setup = false;
		return;
	}
}

// one sample of the lane's stream (or the start of its next channel)
template <class Tables>
J40_DEV void lf_lane_step(LfLane &L, const J40_GLOBAL DevLfTask &t, const LfLaneFrame &F, const Tables &T) {
	if (lf_lane_done(L)) return;
	if (L.setup) { lf_lane_setup(L, t, F); if (L.chan == 7 || L.err) return; }
	lane_bits_refill(L.b);
	const int32_t x = L.x, y = L.y, cw = L.cw;
	// neighbours (j40.h:3965-3990): the raw samples are in pw / pww (this row) and a0..a4 (the row above)
	const int32_t pw = x > 0 ? L.pw : y > 0 ? L.a2 : 0;
	const int32_t pn = y > 0 ? L.a2 : pw;
	const int32_t pnw = x > 0 && y > 0 ? L.a1 : pw;
	const int32_t pne = x + 1 < cw && y > 0 ? L.a3 : pn;
	const int32_t pnee = x + 2 < cw && y > 0 ? L.a4 : pne;
	const int32_t pww = x > 1 ? L.pww : pw;
	const int32_t pnww = x > 1 && y > 0 ? L.a0 : pww;
	int32_t pnn = pn;
	if ((F.uses & 4u) && y > 1) pnn = L.row[x - 2 * cw];
	// the tree walk (j40.h:4181-4216)
	int32_t at = L.root;
	DevTreeNode n = lf_node(F.tree, at);
	while (n.prop >= 0) {
		int32_t val;
		switch (n.prop) {
		case 0: val = L.chan < 3 ? L.chan : L.chan - 3; break;
		case 1: val = L.chan < 3 ? t.sidx0 : t.sidx2; break;
		case 2: val = y; break;
		case 3: val = x; break;
		default: val = neighbour_property(n.prop, x, pw, pn, pnw, pne, pnn, pww, pnww); break;   // 4..14 (props_dev.h; the host admits no other)
		}
		at += val > n.value ? n.a : n.b;
		n = lf_node(F.tree, at);
	}
	uint32_t e2;
	const int32_t u = lane_symbol(L.b, L.state, T, n.value, L.end_bit, &e2);
	int32_t v = unpack_signed_dev(u) * n.b + n.a;
	switch (-1 - n.prop) {   // j40.h:4080
	case 0: break;
	case 1: v += pw; break;
	case 2: v += pn; break;
	case 3: v += (pw + pn) / 2; break;
	case 4: v += mod_abs(pn - pnw) < mod_abs(pw - pnw) ? pw : pn; break;
	case 5: v += mod_gradient(pw, pn, pnw); break;
	case 7: v += pne; break;
	case 8: v += pnw; break;
	case 9: v += pww; break;
	case 10: v += (pw + pnw) / 2; break;
	case 11: v += (pn + pnw) / 2; break;
	case 12: v += (pn + pne) / 2; break;
	default: v += (6 * pn - 2 * pnn + 7 * pw + pww + pnee + 3 * pne + 8) / 16; break;   // 13
	}
	if (e2) { lf_lane_fail(L, e2); return; }
	if (v < -32768 || v > 32767) { lf_lane_fail(L, ERR_POVF); return; }
	L.row[x] = (int16_t) v;
	// move on: the row above slides by one (the sample that enters is x + 3 of the row above; past the row's end it is never looked at)
	L.pww = L.pw; L.pw = v;
	L.a0 = L.a1; L.a1 = L.a2; L.a2 = L.a3; L.a3 = L.a4;
	++L.x;
	if (L.x < cw) {
		// (always two samples ahead of the one that needs it: the load -- an L2 round trip, the row above was written by this lane a
		// few hundred iterations ago -- has two iterations to arrive; loaded just in time it was a third of every iteration)
		if (y > 0 && L.x + 2 < cw) L.a4 = L.row[L.x + 2 - cw];
		return;
	}
	// next row: preload the row above (the row just written) at 0 .. 2
	L.x = 0; ++L.y; L.row += cw; L.pw = L.pww = 0; L.a0 = L.a1 = 0;
	if (L.y < L.chh) {
		L.a2 = L.row[-cw];
		L.a3 = cw > 1 ? L.row[1 - cw] : 0;
		L.a4 = cw > 2 ? L.row[2 - cw] : 0;
		return;
	}
	++L.chan; L.setup = true;
}

} // namespace j40hip
