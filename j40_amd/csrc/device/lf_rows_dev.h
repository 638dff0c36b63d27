// j40_amd/csrc/device/lf_rows_dev.h -- the LfGroup sections of VarDCT frames, one section per wavefront LANE, with everything a
// sample's dependent chain touches held in LDS (k_lf_rows, lf_decode.hip; SURVEY.md 8f-1: the two Modular sub-images of
// j40__lf_group, j40.h:6722-6790 = j40__modular_channel, j40.h:4127-4240, over the LF coefficient image and the HF metadata image).
//
// lf_lanes_dev.h's decoder (same decomposition: a lane is a section, an iteration is one sample of every lane) spent ~2600 cycles
// per sample on ~130 instructions: what it waited for was memory. Its alias entry came from global memory (an L2 round trip on the
// chain of every sample), the row above was loaded from global memory two samples ahead, every sample was a 2-byte global store,
// and on gfx9 one counter (vmcnt) covers loads AND stores: each wait for the alias entry also waited for the store of the sample
// before. Here no sample of a row touches vector memory:
//   * the alias tables of the lane's frame sit in LDS beside its tree (8 bytes per bucket: 14 KB for seven clusters of 256);
//   * a leaf of the staged tree carries its cluster and that cluster's hybrid-integer configuration (context map and configuration
//     table resolved while staging: two dependent LDS reads fewer per sample), and the node a channel's walk starts from is kept
//     in registers -- a channel whose subtree is a single leaf reads nothing but its alias entry;
//   * every lane owns a window of LF_ROW_WIN samples in LDS. The row being decoded is written there; the slot of sample x holds
//     the row ABOVE's sample x until it is overwritten, so the same window serves the neighbours N, NW, NE, NEE, NWW (read two
//     samples ahead of their use, off the chain). A finished row leaves as one coalesced copy done by all 64 lanes together
//     (lf_row_flush_wave): 2-byte stores of consecutive addresses instead of 64 scattered ones per iteration, and nothing waits
//     for them;
//   * channels wider than the window (the varblock-info channel: two rows of up to 65 536 samples) go through the window in
//     pieces and read the row above from global memory like the older decoder did.
// What is left in vector memory: the codestream word requested one refill ahead (once in several samples) and the row copies.
// Same streams as lf_lanes_dev.h (rANS without LZ77, no weighted predictor, no previous-channel properties, plain second header),
// same samples, same status codes: tests/hostsim runs both against the host decoder.
//
// RAW channels (round 6). A channel whose subtree is a single leaf is coded with one context whatever its samples come to, so nothing a
// lane's parse needs depends on the prediction: the lane stores the RESIDUAL (unpacked, times the leaf's multiplier, plus its
// offset) where the sample belongs and skips the neighbours, the prediction and the registers that slide along the row above -- the
// straight-line step without any of that (no lane of the wavefront predicting) is 65 instructions instead of 100-165 --, and
// k_lf_predict (lf_decode.hip) adds the predictions afterwards, a wavefront per section, 64 rows at a time each three columns behind
// the row above (lf_predict_* below: the neighbour rules of j40.h:3965-3990 and the predictors of j40.h:4080 on final samples; the
// first sample that leaves the int16 range is "povf" if it lies before whatever else ended the lane). A residual that does not fit
// the plane's 16 bits says nothing about its sample (the prediction may bring it back): the lane reports ERR_LFFB and the host
// decodes the section, as for any other form the device does not take.
#pragma once
#include "lf_lanes_dev.h"
#include "modular_dev.h"

namespace j40hip {

enum { LF_USES_RAW = 1u << 8 };   // LfRowTables::uses: leaf-only channels leave residuals for k_lf_predict (set by the kernel, not the host's tree scan)
// DevLfResult::raw_mask: four bits per channel 0..6, predictor + 1 of a channel left as residuals (0: the channel holds samples);
// DevLfResult::stopped_at: channel << 24 | samples of that channel completed, where the lane ended (7 << 24: all of the section)
J40_DEV uint32_t lf_stopped_at(int32_t chan, int32_t done) { return ((uint32_t) chan << 24) | (uint32_t) done; }

enum { LF_ROW_WIN = 256,             // samples per lane window: an LfGroup is at most 256 cells wide (2048 pixels)
       LF_ROW_PITCH = LF_ROW_WIN + 2 };   // int16 units between the windows of neighbouring lanes (129 dwords: lanes at the same x hit distinct banks)

// the tables of a lane's frame as k_lf_rows stages them
struct LfRowTables {
	const J40_LDS DevTreeNode *tree;   // leaves: value = cluster << 24 | configuration word of the cluster (lf_rows_leaf_word)
	const J40_LDS uint64_t *alias;
	int32_t log_alpha, log_bucket;
	uint32_t uses;                     // LfLaneFrame::uses
};

// what a staged leaf carries instead of its context: `cfg` as in LaneTables::cluster_cfg (max_token from bit 12 up; tokens are
// < 256, so clamping it to 11 bits keeps `token > max_token` intact), the cluster in the top byte, and bit 23 when a token of the
// cluster can ask for more than 17 extra bits (a second refill inside the symbol: not for the straight-line step)
enum { LF_LEAF_CFG_MASK = 0x7fffff, LF_LEAF_WIDE = 1 << 23 };
J40_DEV int32_t lf_rows_leaf_word(uint32_t cluster, uint32_t cfg) {
	const uint32_t mt = cfg >> 12, split_exp = cfg & 15u, in_token = ((cfg >> 4) & 15u) + ((cfg >> 8) & 15u);
	const uint32_t top_token = mt > 255u ? 255u : mt, split = 1u << split_exp;
	const uint32_t most_extra = top_token >= split ? split_exp - in_token + ((top_token - split) >> in_token) : 0u;   // (split_exp >= in_token: j40.h:2313)
	return (int32_t) ((cfg & 0xfffu) | ((mt > 0x7ffu ? 0x7ffu : mt) << 12) | (most_extra > 17u ? (uint32_t) LF_LEAF_WIDE : 0u) | (cluster << 24));
}

// LDS bytes of one frame's staged tables (tree nodes, then the alias tables)
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t lf_rows_table_bytes(int32_t num_nodes, int32_t num_clusters, int32_t log_alpha) {
	return ((16u * (uint32_t) num_nodes + 15u) & ~15u) + 8u * ((uint32_t) num_clusters << log_alpha);
}

struct LfRowLane {
	LaneBits b;
	uint32_t state, err, end_bit;
	int32_t chan;                      // as LfLane::chan
	int32_t x, y, cw, chh;
	int32_t r_prop, r_value, r_a, r_b, root;   // the node the channel's walk starts from, and where it is
	int32_t pw, pww;
	int32_t a0, a1, a2, a3, a4;        // the row above at x - 2 .. x + 2
	J40_GLOBAL int16_t *row;           // the channel's current row in global memory
	J40_LDS int16_t *win;              // this lane's window
	int32_t nb_varblocks;
	bool setup;
	int32_t flush_n;                   // > 0: the step completed a piece of a row: win[0 .. flush_n) belongs at flush_dst
	J40_GLOBAL int16_t *flush_dst;
	// the channel in the form lf_row_step_plain takes (lf_row_plan_channel): its subtree is one leaf, or one test over two leaves
	// that predict alike
	int32_t plain_left;                // how many of the lane's next samples take the straight-line step (set by the general step)
	bool live;                         // not finished (lf_row_done), as of the lane's last general step
	bool raw;                          // the channel is left as residuals (the header of this file)
	uint32_t raw_mask, stopped_at;     // DevLfResult's words
	bool plain_ok, plain_wide;         // plain_wide: rows wider than the window, and nothing of the channel looks at the row above
	int32_t k_thr; uint32_t k_word_gt, k_word_le;            // the test's threshold; the leaf words behind "greater" and "not greater"
	int32_t c_x, c_y, c_w, c_n, c_nw, c_ne, c_ww, c_nww;     // the tested property as a signed sum of position and neighbours ...
	bool c_abs, c_first;                                     // ... its magnitude (4, 5); W itself in the first column (8)
	int32_t p_kind, p_w, p_n, p_nw, p_ne, p_ww; bool p_half; // the prediction: 0 a sum of neighbours (halved: the averages), 1 "select", 2 the clamped gradient
	int32_t p_mul, p_off;                                    // the leaves' multiplier and offset
};

// (where a lane ends: the channel it was in and how many of its samples are complete -- x, y still say so)
J40_DEV void lf_row_fail(LfRowLane &L, uint32_t e) {
	if (!L.err) L.err = e;
	if (L.chan < 7) L.stopped_at = lf_stopped_at(L.chan, L.setup ? 0 : L.y * L.cw + L.x);
	L.chan = 7; L.setup = false;
}
J40_DEV bool lf_row_done(const LfRowLane &L) { return L.chan == 7 && !L.setup; }

J40_DEV uint32_t lf_row_take(LfRowLane &L, int32_t n) {   // header bits (n <= 31)
	if (L.b.nbits < n) lane_bits_refill(L.b);
	const uint32_t v = lane_bits_take(L.b, n);
	if (lane_bit_position(L.b) > L.end_bit) lf_row_fail(L, ERR_SHRT);
	return v;
}

J40_DEV void lf_row_finish_code(LfRowLane &L) {   // j40.h:2884
	if (L.state == 0) { lane_bits_refill(L.b); L.state = lf_row_take(L, 16); L.state |= lf_row_take(L, 16) << 16; }
	if (!L.err && L.state != 0x130000) lf_row_fail(L, ERR_ANS);
	L.state = 0;
}

J40_DEV void lf_row_init(LfRowLane &L, const J40_GLOBAL DevLfTask &t, J40_LDS int16_t *win) {
	lane_bits_init(L.b, (const J40_GLOBAL uint8_t *) t.codestream, 8u * t.byte_off + t.bit_off);
	L.state = 0; L.err = 0; L.end_bit = 8u * (t.byte_off + t.size);
	L.chan = 0; L.setup = true; L.nb_varblocks = 0;
	L.x = L.y = 0; L.cw = L.chh = 0; L.root = 0; L.r_prop = -1; L.r_value = L.r_a = L.r_b = 0;
	L.pw = L.pww = 0; L.a0 = L.a1 = L.a2 = L.a3 = L.a4 = 0; L.row = nullptr; L.win = win;
	L.flush_n = 0; L.flush_dst = nullptr;
	L.plain_left = 0; L.live = true;
	L.raw = false; L.raw_mask = 0; L.stopped_at = lf_stopped_at(7, 0);
	L.plain_ok = L.plain_wide = false; L.k_thr = 0; L.k_word_gt = L.k_word_le = 0;
	L.c_x = L.c_y = L.c_w = L.c_n = L.c_nw = L.c_ne = L.c_ww = L.c_nww = 0; L.c_abs = L.c_first = false;
	L.p_kind = 0; L.p_w = L.p_n = L.p_nw = L.p_ne = L.p_ww = 0; L.p_half = false; L.p_mul = 1; L.p_off = 0;
	if (8u * t.byte_off + t.bit_off > L.end_bit) lf_row_fail(L, ERR_SHRT);
}

// Can the channel that starts at L.root go through lf_row_step_plain? Its subtree has to be a single leaf, or ONE test of a
// property of the sample's position / neighbourhood (2..12, 14: no test that needs the row two up) over two leaves with the same
// predictor, offset and multiplier -- what encoders write for these channels: libjxl's LF trees test the gradient property and
// predict with the clamped gradient throughout --, the predictor one of 0..5, 7..12, the rows no wider than the window, and no
// token of the leaves' clusters may ask for a second refill. Fills in the channel's constants; any other channel keeps the general step.
J40_DEV void lf_row_plan_channel(LfRowLane &L, const LfRowTables &T) {
	L.plain_ok = L.plain_wide = false; L.raw = false;
	DevTreeNode leaf;
	leaf.prop = L.r_prop; leaf.value = L.r_value; leaf.a = L.r_a; leaf.b = L.r_b;
	L.c_x = L.c_y = L.c_w = L.c_n = L.c_nw = L.c_ne = L.c_ww = L.c_nww = 0; L.c_abs = L.c_first = false; L.k_thr = 0;
	if (L.r_prop >= 0) {
		const DevTreeNode gt = lf_node(T.tree, L.root + L.r_a), le = lf_node(T.tree, L.root + L.r_b);
		if (gt.prop >= 0 || le.prop != gt.prop || le.a != gt.a || le.b != gt.b) return;
		switch (L.r_prop) {
		case 2: L.c_y = 1; break;
		case 3: L.c_x = 1; break;
		case 4: L.c_n = 1; L.c_abs = true; break;
		case 5: L.c_w = 1; L.c_abs = true; break;
		case 6: L.c_n = 1; break;
		case 7: L.c_w = 1; break;
		case 8: L.c_w = 1; L.c_ww = -1; L.c_nw = -1; L.c_nww = 1; L.c_first = true; break;
		case 9: L.c_w = 1; L.c_n = 1; L.c_nw = -1; break;
		case 10: L.c_w = 1; L.c_nw = -1; break;
		case 11: L.c_nw = 1; L.c_n = -1; break;
		case 12: L.c_n = 1; L.c_ne = -1; break;
		case 14: L.c_w = 1; L.c_ww = -1; break;
		default: return;   // 13 looks two rows up (0 and 1 were decided when the channel started)
		}
		L.k_thr = L.r_value; L.k_word_gt = (uint32_t) gt.value; L.k_word_le = (uint32_t) le.value;
		leaf = gt;
	} else L.k_word_gt = L.k_word_le = (uint32_t) L.r_value;
	if ((L.k_word_gt | L.k_word_le) & (uint32_t) LF_LEAF_WIDE) return;
	L.p_kind = 0; L.p_w = L.p_n = L.p_nw = L.p_ne = L.p_ww = 0; L.p_half = false;
	switch (-1 - leaf.prop) {   // j40.h:4080
	case 0: break;
	case 1: L.p_w = 1; break;
	case 2: L.p_n = 1; break;
	case 3: L.p_w = 1; L.p_n = 1; L.p_half = true; break;
	case 4: L.p_kind = 1; break;
	case 5: L.p_kind = 2; break;
	case 7: L.p_ne = 1; break;
	case 8: L.p_nw = 1; break;
	case 9: L.p_ww = 1; break;
	case 10: L.p_w = 1; L.p_nw = 1; L.p_half = true; break;
	case 11: L.p_n = 1; L.p_nw = 1; L.p_half = true; break;
	case 12: L.p_n = 1; L.p_ne = 1; L.p_half = true; break;
	default: return;   // 13 (the weighted average looks two rows up and two to the right)
	}
	L.p_off = leaf.a; L.p_mul = leaf.b;
	if (L.cw > LF_ROW_WIN) {
		// rows wider than the window (the varblock-info channel) keep the row above in global memory: plain only when neither the test
		// nor the prediction looks up -- W and WW travel in registers -- and then not for a row's first sample (W falls back to N there)
		if (L.c_n | L.c_nw | L.c_ne | L.c_nww | L.p_n | L.p_nw | L.p_ne | L.p_kind) return;
		L.plain_wide = true;
	}
	L.plain_ok = true;
	// a single leaf: residuals now, predictions later -- rows wider than the window only when there is nothing to predict (a wide
	// row's prediction would be one lane's walk along it)
	const int32_t predictor = -1 - leaf.prop;
	if ((T.uses & (uint32_t) LF_USES_RAW) && L.r_prop < 0 && (!L.plain_wide || predictor == 0)) {
		L.raw = true;
		L.raw_mask |= (uint32_t) (predictor + 1) << (4 * L.chan);
	}
}

// starts channel L.chan: as lf_lane_setup, and fetches the node the channel's walk starts from
J40_DEV void lf_row_setup(LfRowLane &L, const J40_GLOBAL DevLfTask &t, const LfRowTables &T) {
	for (;;) {
		if (L.chan == 3) {
			lf_row_finish_code(L);
			if (L.err) return;
			L.nb_varblocks = (int32_t) lf_row_take(L, t.nbvb_bits) + 1;
			const uint32_t header = lf_row_take(L, 4);   // use_global_tree = 1, default wp = 1, no transforms (j40.h:3717-3760)
			if (L.err) return;
			if (header != 3u || 2u * (uint32_t) L.nb_varblocks > t.info_capacity) { lf_row_fail(L, ERR_LFFB); return; }
		}
		if (L.chan == 7) { lf_row_finish_code(L); L.chan = 7; L.setup = false; return; }
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
		const int32_t cidx = L.chan < 3 ? L.chan : L.chan - 3, sidx = L.chan < 3 ? t.sidx0 : t.sidx2;
		int32_t at = 0;
		DevTreeNode n;
		for (;;) {
			n = lf_node(T.tree, at);
			if (n.prop == 0) at += cidx > n.value ? n.a : n.b;
			else if (n.prop == 1) at += sidx > n.value ? n.a : n.b;
			else break;
		}
		L.root = at; L.r_prop = n.prop; L.r_value = n.value; L.r_a = n.a; L.r_b = n.b;
		lf_row_plan_channel(L, T);
		L.setup = false;
		return;
	}
}

// one sample of the lane's stream (or the start of its next channel). The caller copies a completed piece out (L.flush_n) before
// the lane's next step.
J40_DEV void lf_row_step_general(LfRowLane &L, const J40_GLOBAL DevLfTask &t, const LfRowTables &T) {
	if (lf_row_done(L)) return;
	if (L.err) {   // the straight-line step raised it (lf_plain_commit moved x past the sample): the lane ends here, x - 1 samples into its row
		L.x -= 1;
		if (L.raw && !L.plain_wide && L.x > 0) { L.flush_n = L.x; L.flush_dst = L.row; }   // (k_lf_predict looks at the samples before it)
		lf_row_fail(L, L.err);
		return;
	}
	if (L.setup) { lf_row_setup(L, t, T); if (L.chan == 7 || L.err) return; }
	lane_bits_refill(L.b);
	const int32_t x = L.x, y = L.y, cw = L.cw;
	const bool wide = cw > LF_ROW_WIN;
	// neighbours (j40.h:3965-3990)
	const int32_t pw = x > 0 ? L.pw : y > 0 ? L.a2 : 0;
	const int32_t pn = y > 0 ? L.a2 : pw;
	const int32_t pnw = x > 0 && y > 0 ? L.a1 : pw;
	const int32_t pne = x + 1 < cw && y > 0 ? L.a3 : pn;
	const int32_t pnee = x + 2 < cw && y > 0 ? L.a4 : pne;
	const int32_t pww = x > 1 ? L.pww : pw;
	const int32_t pnww = x > 1 && y > 0 ? L.a0 : pww;
	int32_t pnn = pn;
	if ((T.uses & 4u) && y > 1) pnn = L.row[x - 2 * cw];   // (two rows up: copied out a row ago)
	// the tree walk (j40.h:4181-4216) from the channel's node
	DevTreeNode n;
	n.prop = L.r_prop; n.value = L.r_value; n.a = L.r_a; n.b = L.r_b;
	int32_t at = L.root;
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
		n = lf_node(T.tree, at);
	}
	uint32_t e2;
	const uint32_t word = (uint32_t) n.value;
	const int32_t u = lane_symbol_in_cluster(L.b, L.state, T.alias, T.log_alpha, T.log_bucket, word >> 24, word & (uint32_t) LF_LEAF_CFG_MASK, L.end_bit, &e2);
	int32_t v = unpack_signed_dev(u) * n.b + n.a;
	if (!L.raw) switch (-1 - n.prop) {   // j40.h:4080
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
	if (e2 || v < -32768 || v > 32767) {
		if (L.raw && !wide && x > 0) { L.flush_n = x; L.flush_dst = L.row; }
		lf_row_fail(L, e2 ? e2 : L.raw ? (uint32_t) ERR_LFFB : (uint32_t) ERR_POVF);
		return;
	}
	L.win[x & (LF_ROW_WIN - 1)] = (int16_t) v;
	L.pww = L.pw; L.pw = v;
	L.a0 = L.a1; L.a1 = L.a2; L.a2 = L.a3; L.a3 = L.a4;
	const int32_t nx = x + 1;
	L.x = nx;
	if (nx < cw) {
		// the sample that enters the registers is x + 3 of the row above: still in its slot of the window (this row has reached
		// x), or -- rows wider than the window -- in the row as copied out
		if (y > 0 && nx + 2 < cw) L.a4 = wide ? (int32_t) L.row[nx + 2 - cw] : (int32_t) L.win[nx + 2];
		if (wide && (nx & (LF_ROW_WIN - 1)) == 0) { L.flush_n = LF_ROW_WIN; L.flush_dst = L.row + (nx - LF_ROW_WIN); }
		return;
	}
	// the row is complete: its last piece goes out; the next row finds this one in the window (or in global memory)
	L.flush_n = ((cw - 1) & (LF_ROW_WIN - 1)) + 1; L.flush_dst = L.row + (cw - L.flush_n);
	L.x = 0; L.y = y + 1; L.row += cw; L.pw = L.pww = 0; L.a0 = L.a1 = 0;
	if (L.y < L.chh) {
		if (wide) { L.a2 = L.row[-cw]; L.a3 = L.row[1 - cw]; L.a4 = L.row[2 - cw]; }   // (cw > 256)
		else { L.a2 = L.win[0]; L.a3 = cw > 1 ? L.win[1] : 0; L.a4 = cw > 2 ? L.win[2] : 0; }
		return;
	}
	++L.chan; L.setup = true;
}

// ---- the straight-line step ----
// Measured on an MI355X (tools/ubench/lone_wave.hip), a wavefront alone on its SIMD: a dependent vector instruction every 8.6
// cycles (5.3 with four independent chains), an LDS read 56-67, an L1-hit global load 200 -- and SIXTY cycles for every `if` the
// wavefront walks past (compare, s_and_saveexec, s_cbranch_execz, s_or), whether its body runs or not. The general step above is
// some fifty of those per sample (the switches over properties and predictors, refills, edges, errors): 3 200 cycles per sample
// whatever its memory accesses cost, which is why moving them to LDS alone changed nothing (434 ms per launch against 410).
// lf_row_step_plain is the same sample as ONE basic block: the property is a signed sum with per-channel coefficients, the tree
// walk a compare and a select, the refill, the renormalisation, the extra bits, the prediction (a sum, "select" and the clamped
// gradient all computed, one chosen) and the error bookkeeping are selects. A lane takes it when its channel has the form
// lf_row_plan_channel accepts, its first symbol has been read and the sample is not the last of its row; everything else -- channel
// starts, row ends, other trees -- is the general step, which the kernel runs for the lanes that need it after the others' plain
// step. The general step leaves the number of plain samples that follow (plain_left), so that the choice costs one compare.
// how many samples from here on are plain ones: to the last but one of the row (the last one ends the row: the general step), in
// wide rows from the second sample to the one before the next piece of the window is complete
J40_DEV int32_t lf_row_plain_run(const LfRowLane &L) {
	const bool can = L.plain_ok & !L.setup & (L.chan < 7) & (L.state != 0) & (!L.plain_wide | (L.x > 0));
	const int32_t last = L.plain_wide ? mod_min(L.cw - 1, L.x | (LF_ROW_WIN - 1)) : L.cw - 1;   // the first sample that is not plain
	return can ? mod_max(last - L.x, 0) : 0;
}
// the step the kernel gives the lanes that are not in a run of plain samples; leaves the length of the run that follows
J40_DEV void lf_row_step(LfRowLane &L, const J40_GLOBAL DevLfTask &t, const LfRowTables &T) {
	lf_row_step_general(L, t, T);
	L.plain_left = lf_row_plain_run(L);
	L.live = !lf_row_done(L);
}

// a * b + c for a coefficient -1 .. 1 and a sample or position within 17 bits: v_mad_i32_i24, one full-rate instruction (the
// compiler turns __mul24 into a sign extension and a quarter-rate 32-bit multiply here)
#ifdef __HIPCC__
J40_DEV int32_t lf_mad24(int32_t a, int32_t b, int32_t c) { int32_t d; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
#else
J40_DEV int32_t lf_mad24(int32_t a, int32_t b, int32_t c) { return a * b + c; }
#endif
// unpack_signed_dev for the token values a symbol without an error yields (0 <= u < 2^30), as two instructions instead of a branch
J40_DEV int32_t lf_unzigzag(int32_t u) { return (int32_t) ((uint32_t) u >> 1) ^ -(u & 1); }

// The step in three phases, so that TWO sections can go through it side by side on one lane (lf_row_step_plain2, k_lf_rows<true>):
// everything that reads LDS comes before anything that writes it, and between the phases the two sections' instructions are
// independent of each other; compiled with the max-ILP scheduling strategy they do come out interleaved. MEASURED AND NOT THE DEFAULT:
// a launch of 128 8K frames takes 421-450 ms with two sections per lane against 212-222 ms with one -- the same cycles per sample.
// The step runs at about nine cycles per instruction either way (1 690 cycles for its 185, mostly 8-byte encodings); the
// microbenchmark's best, four independent chains of 4-byte instructions, was 5.3. What bounds a lone wavefront here is not the
// dependence between its instructions but how fast it is fed them, and only fewer (or shorter) instructions per sample help.
// What the lanes of a wavefront need of the step, taken together (wave-uniform; lf_plain_needs, recomputed whenever a lane has been
// through the general step): the step is instantiated for every combination and the kernel jumps to the one that covers its lanes --
// a wavefront whose lanes all sit in leaf-only channels predicted by the clamped gradient runs 150 instructions a sample, not 185.
enum { LF_NEED_TEST = 1,    // some lane's channel has a test (two different leaf words)
       LF_NEED_LIN = 2, LF_NEED_SEL = 4, LF_NEED_GRAD = 8,   // the predictions some lane uses: a (halved) sum of neighbours, "select", the clamped gradient
       LF_NEED_MUL = 16,    // some lane's leaves have a multiplier other than 1 or an offset
       LF_NEED_ALL = 31,
       LF_NEED_PRED = LF_NEED_TEST | LF_NEED_LIN | LF_NEED_SEL | LF_NEED_GRAD };   // none of these: every lane of the step is in a RAW channel

struct LfPlainCtx {
	int32_t x, slot, ahead3, pw, pn, pnw, pne, pww;
	uint32_t word, idx, bucket; uint64_t entry;
};

// refill, neighbours, the property and the walk; asks LDS for the alias entry and the row above's next sample
template <uint32_t NEED>
J40_DEV void lf_plain_front(LfRowLane &L, const LfRowTables &T, LfPlainCtx &c) {
	LaneBits &b = L.b;
	{   // lane_bits_refill as selects; the word after next is asked for every time (the same word again when nothing was appended)
		const bool need = b.nbits <= 32;
		b.bits |= (uint64_t) (need ? b.ahead : 0u) << (need ? b.nbits : 0);
		b.nbits += need ? 32 : 0; b.pos += need ? 4u : 0u;
		b.ahead = lane_load32(b.base, b.pos);
	}
	const int32_t x = L.x, y = L.y, cw = L.cw;
	c.x = x; c.slot = x & (LF_ROW_WIN - 1);   // (= x unless the row is wider than the window)
	if (!(NEED & LF_NEED_PRED)) {   // residuals only: no neighbour is looked at
		c.ahead3 = 0; c.pw = c.pn = c.pnw = c.pne = c.pww = 0;
		c.word = L.k_word_le;
		c.idx = L.state & 0xfff; c.bucket = c.idx >> T.log_bucket;
		c.entry = T.alias[((c.word >> 24) << T.log_alpha) + c.bucket];
		return;
	}
	c.ahead3 = L.win[c.slot + 3];   // the row above at x + 3, for the next sample's registers (slots past the row's end are never looked at)
	const bool up = y > 0, left = x > 0;
	const int32_t pw = left ? L.pw : up ? L.a2 : 0;
	const int32_t pn = up ? L.a2 : pw;
	const int32_t pnw = left && up ? L.a1 : pw;
	const int32_t pne = x + 1 < cw && up ? L.a3 : pn;
	const int32_t pww = x > 1 ? L.pww : pw;
	const int32_t pnww = x > 1 && up ? L.a0 : pww;
	c.pw = pw; c.pn = pn; c.pnw = pnw; c.pne = pne; c.pww = pww;
	// the property and the walk: one test, or none (both words the leaf's)
	if (NEED & LF_NEED_TEST) {
		int32_t val = lf_mad24(L.c_x, x, lf_mad24(L.c_y, y, lf_mad24(L.c_w, pw, lf_mad24(L.c_n, pn, 0)))) + lf_mad24(L.c_nw, pnw, lf_mad24(L.c_ne, pne, lf_mad24(L.c_ww, pww, lf_mad24(L.c_nww, pnww, 0))));   // (two chains of four)
		val = L.c_abs ? mod_abs(val) : val;
		val = L.c_first && !left ? pw : val;
		c.word = val > L.k_thr ? L.k_word_gt : L.k_word_le;
	} else c.word = L.k_word_le;   // (no lane tests anything: both words are the leaf's)
	// the alias entry of the state's bucket in the leaf's cluster (lane_symbol_in_cluster)
	c.idx = L.state & 0xfff; c.bucket = c.idx >> T.log_bucket;
	c.entry = T.alias[((c.word >> 24) << T.log_alpha) + c.bucket];
}

// the symbol (rANS step + hybrid integer, lane_symbol_in_cluster<STRAIGHT> from the alias entry on), the prediction, the sample;
// returns the sample, *code = the error this sample raises (0: none)
template <uint32_t NEED>
J40_DEV int32_t lf_plain_middle(LfRowLane &L, const LfRowTables &T, const LfPlainCtx &c, uint32_t *code) {
	LaneBits &b = L.b;
	const uint32_t m = c.word & (uint32_t) LF_LEAF_CFG_MASK, pos = c.idx & ((1u << T.log_bucket) - 1);
	const uint32_t elo = (uint32_t) c.entry, ehi = (uint32_t) (c.entry >> 32);
	const bool aliased = pos >= (elo & 0xff);
	const int32_t token = (int32_t) (aliased ? (elo >> 20) & 0xff : c.bucket);
	const uint32_t offset = aliased ? (elo >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (c.entry >> 28) & 0x1fff : (ehi >> 9) & 0x1fff;
	uint32_t state = d * (L.state >> 12) + offset + pos;
	const bool renorm = state < (1u << 16);
	const uint32_t low = lane_bits_take(b, renorm ? 16 : 0);
	L.state = renorm ? (state << 16) | low : state;
	const bool short1 = lane_bit_position(b) > L.end_bit;
	const int32_t split_exp = (int32_t) (m & 15), split = 1 << split_exp;
	const bool big = token >= split;
	const int32_t mt = (int32_t) (m >> 12);
	const bool iovf = big && token > mt;
	const int32_t tok = iovf ? mt : token;
	const int32_t msb = (int32_t) ((m >> 4) & 15), lsb = (int32_t) ((m >> 8) & 15), in_token = msb + lsb;
	const int32_t midbits = big ? split_exp - in_token + ((tok - split) >> in_token) : 0;   // (<= 17: lf_rows_leaf_word; the window holds them)
	const int32_t mid = (int32_t) lane_bits_take(b, midbits);
	const bool short2 = lane_bit_position(b) > L.end_bit;
	const int32_t top = 1 << msb;
	const int32_t lo = tok & ((1 << lsb) - 1), hi = (tok >> lsb) & (top - 1);
	const int32_t value = ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
	const uint32_t e2 = short1 ? (uint32_t) ERR_SHRT : iovf ? (uint32_t) ERR_IOVF : short2 ? (uint32_t) ERR_SHRT : 0u;
	const int32_t u = big ? value : token;
	// the prediction
	int32_t pred = 0;
	if (NEED & LF_NEED_LIN) {
		int32_t lin = lf_mad24(L.p_w, c.pw, lf_mad24(L.p_n, c.pn, 0)) + lf_mad24(L.p_nw, c.pnw, lf_mad24(L.p_ne, c.pne, lf_mad24(L.p_ww, c.pww, 0)));
		pred = L.p_half ? (lin + (int32_t) ((uint32_t) lin >> 31)) >> 1 : lin;   // (a + b) / 2, towards zero
	}
	if (NEED & LF_NEED_SEL) {
		const int32_t sel = mod_abs(c.pn - c.pnw) < mod_abs(c.pw - c.pnw) ? c.pw : c.pn;
		pred = (NEED & (LF_NEED_LIN | LF_NEED_GRAD)) ? (L.p_kind == 1 ? sel : pred) : sel;   // (the only kind around: no lane to tell apart)
	}
	if (NEED & LF_NEED_GRAD) {
		const int32_t grad = mod_gradient(c.pw, c.pn, c.pnw);
		pred = (NEED & (LF_NEED_LIN | LF_NEED_SEL)) ? (L.p_kind == 2 ? grad : pred) : grad;
	}
	const int32_t res = (NEED & LF_NEED_MUL) ? lf_unzigzag(u) * L.p_mul + L.p_off : lf_unzigzag(u);
	const int32_t v = (NEED & LF_NEED_PRED) ? res + (L.raw ? 0 : pred) : res;
	const uint32_t range = (NEED & LF_NEED_PRED) ? (L.raw ? (uint32_t) ERR_LFFB : (uint32_t) ERR_POVF) : (uint32_t) ERR_LFFB;   // (a residual too wide for the plane: the header of this file)
	*code = e2 ? e2 : v < -32768 || v > 32767 ? range : 0u;
	return v;
}

// an error ends the lane's run (the general step it goes to next ends the lane and notes where: what the failing sample left in the
// window is never looked at); otherwise the sample is stored and the registers slide
template <uint32_t NEED>
J40_DEV void lf_plain_commit(LfRowLane &L, const LfPlainCtx &c, int32_t v, uint32_t code) {
	L.err = L.err ? L.err : code;
	L.plain_left = code ? 0 : L.plain_left - 1;
	L.win[c.slot] = (int16_t) v;
	if (NEED & LF_NEED_PRED) {
		L.pww = L.pw; L.pw = v;
		L.a0 = L.a1; L.a1 = L.a2; L.a2 = L.a3; L.a3 = L.a4; L.a4 = c.ahead3;
	}
	L.x = c.x + 1;
}

template <uint32_t NEED>
J40_DEV void lf_row_step_plain_for(LfRowLane &L, const LfRowTables &T) {
	LfPlainCtx c; uint32_t code;
	lf_plain_front<NEED>(L, T, c);
	const int32_t v = lf_plain_middle<NEED>(L, T, c, &code);
	lf_plain_commit<NEED>(L, c, v, code);
}
J40_DEV void lf_row_step_plain(LfRowLane &L, const LfRowTables &T) { lf_row_step_plain_for<LF_NEED_ALL>(L, T); }

// this lane's share of the wavefront's needs (0 for a lane that takes no plain step)
J40_DEV uint32_t lf_plain_needs(const LfRowLane &L) {
	if (!(L.live & L.plain_ok)) return 0;
	if (L.raw) return L.p_mul != 1 || L.p_off != 0 ? (uint32_t) LF_NEED_MUL : 0u;   // (nothing of the prediction)
	return (L.k_word_gt != L.k_word_le ? (uint32_t) LF_NEED_TEST : 0u) | (L.p_kind == 0 ? (uint32_t) LF_NEED_LIN : L.p_kind == 1 ? (uint32_t) LF_NEED_SEL : (uint32_t) LF_NEED_GRAD)
		| (L.p_mul != 1 || L.p_off != 0 ? (uint32_t) LF_NEED_MUL : 0u);
}
// the step for lanes whose needs add up to `need`
#define LF_PLAIN_CASE(n) case n: lf_row_step_plain_for<n>(L, T); break;
J40_DEV void lf_row_step_plain_needs(LfRowLane &L, const LfRowTables &T, uint32_t need) {
	switch (need) {
	LF_PLAIN_CASE(0) LF_PLAIN_CASE(1) LF_PLAIN_CASE(2) LF_PLAIN_CASE(3) LF_PLAIN_CASE(4) LF_PLAIN_CASE(5) LF_PLAIN_CASE(6) LF_PLAIN_CASE(7)
	LF_PLAIN_CASE(8) LF_PLAIN_CASE(9) LF_PLAIN_CASE(10) LF_PLAIN_CASE(11) LF_PLAIN_CASE(12) LF_PLAIN_CASE(13) LF_PLAIN_CASE(14) LF_PLAIN_CASE(15)
	LF_PLAIN_CASE(16) LF_PLAIN_CASE(17) LF_PLAIN_CASE(18) LF_PLAIN_CASE(19) LF_PLAIN_CASE(20) LF_PLAIN_CASE(21) LF_PLAIN_CASE(22) LF_PLAIN_CASE(23)
	LF_PLAIN_CASE(24) LF_PLAIN_CASE(25) LF_PLAIN_CASE(26) LF_PLAIN_CASE(27) LF_PLAIN_CASE(28) LF_PLAIN_CASE(29) LF_PLAIN_CASE(30)
	default: lf_row_step_plain_for<LF_NEED_ALL>(L, T); break;
	}
}
#undef LF_PLAIN_CASE
#ifdef __HIPCC__
// The kernel's inner loop: the wavefront's lanes step through their runs of plain samples until some live lane is out of its run
// (a channel start, a row's end: the caller's business). The choice among the instantiations is made once per such stretch -- a
// wave-uniform switch is a tree of scalar branches, too dear to walk per sample --, a stretch being hundreds of samples.
template <uint32_t NEED>
J40_DEV void lf_row_run_plain_for(LfRowLane &L, const LfRowTables &T) {
	// (Measured in round 6: the same loop with the execution mask set once per stretch -- `if (plain) do step while (all still plain)`,
	// the loop's own test a comparison of two scalars -- is SLOWER, 151 ms per launch against 135: profiles/r06_lf_rows_*_call_h*.)
	for (;;) {   // (the lanes in a run step at least once per call: a lane that stays in a channel of another form does not hold them up)
		const bool plain = L.plain_left > 0;
		if (!__builtin_amdgcn_ballot_w64(plain)) return;
		if (plain) lf_row_step_plain_for<NEED>(L, T);
		if (__builtin_amdgcn_ballot_w64(!(L.plain_left > 0) & L.live)) return;
	}
}
#define LF_PLAIN_CASE(n) case n: lf_row_run_plain_for<n>(L, T); break;
J40_DEV void lf_row_run_plain(LfRowLane &L, const LfRowTables &T, uint32_t need) {
	switch (need) {
	LF_PLAIN_CASE(0) LF_PLAIN_CASE(1) LF_PLAIN_CASE(2) LF_PLAIN_CASE(3) LF_PLAIN_CASE(4) LF_PLAIN_CASE(5) LF_PLAIN_CASE(6) LF_PLAIN_CASE(7)
	LF_PLAIN_CASE(8) LF_PLAIN_CASE(9) LF_PLAIN_CASE(10) LF_PLAIN_CASE(11) LF_PLAIN_CASE(12) LF_PLAIN_CASE(13) LF_PLAIN_CASE(14) LF_PLAIN_CASE(15)
	LF_PLAIN_CASE(16) LF_PLAIN_CASE(17) LF_PLAIN_CASE(18) LF_PLAIN_CASE(19) LF_PLAIN_CASE(20) LF_PLAIN_CASE(21) LF_PLAIN_CASE(22) LF_PLAIN_CASE(23)
	LF_PLAIN_CASE(24) LF_PLAIN_CASE(25) LF_PLAIN_CASE(26) LF_PLAIN_CASE(27) LF_PLAIN_CASE(28) LF_PLAIN_CASE(29) LF_PLAIN_CASE(30)
	default: lf_row_run_plain_for<LF_NEED_ALL>(L, T); break;
	}
}
#endif
#undef LF_PLAIN_CASE

// two sections (each with the tables of its frame), one sample each
J40_DEV void lf_row_step_plain2(LfRowLane &A, LfRowLane &B, const LfRowTables &TA, const LfRowTables &TB) {
	LfPlainCtx ca, cb; uint32_t code_a, code_b;
	lf_plain_front<LF_NEED_ALL>(A, TA, ca);
	lf_plain_front<LF_NEED_ALL>(B, TB, cb);
	const int32_t va = lf_plain_middle<LF_NEED_ALL>(A, TA, ca, &code_a);
	const int32_t vb = lf_plain_middle<LF_NEED_ALL>(B, TB, cb, &code_b);
	lf_plain_commit<LF_NEED_ALL>(A, ca, va, code_a);
	lf_plain_commit<LF_NEED_ALL>(B, cb, vb, code_b);
}

// ---- the predictions of RAW channels (k_lf_predict; tests/hostsim runs the serial form) ----
// where channel `chan` of a section lies and how large it is (as lf_row_setup has it)
J40_DEV J40_GLOBAL int16_t *lf_channel_plane(const J40_GLOBAL DevLfTask &t, int32_t chan, int32_t nb_varblocks, int32_t *cw, int32_t *chh) {
	switch (chan) {
	case 0: case 1: case 2: *cw = t.w8; *chh = t.h8; return (J40_GLOBAL int16_t *) t.lf[chan];
	case 3: *cw = t.w64; *chh = t.h64; return (J40_GLOBAL int16_t *) t.xfromy;
	case 4: *cw = t.w64; *chh = t.h64; return (J40_GLOBAL int16_t *) t.bfromy;
	case 5: *cw = nb_varblocks; *chh = 2; return (J40_GLOBAL int16_t *) t.info;
	default: *cw = t.w8; *chh = t.h8; return (J40_GLOBAL int16_t *) t.sharp;
	}
}
// how many samples of channel `chan` the lane completed, given where it stopped
J40_DEV int32_t lf_channel_complete(uint32_t stopped_at, int32_t chan, int32_t cw, int32_t chh) {
	const int32_t at_chan = (int32_t) (stopped_at >> 24), done = (int32_t) (stopped_at & 0xffffffu);
	return chan < at_chan ? cw * chh : chan == at_chan ? done : 0;
}
// one sample: residual + prediction from the (final) neighbours, edge rules of j40.h:3965-3990
J40_DEV int32_t lf_predict_value(int32_t res, int32_t predictor, int32_t x, int32_t y, int32_t cw, int32_t w, int32_t ww, int32_t nww, int32_t nw, int32_t n, int32_t ne) {
	ModNeigh p;
	p.w = x > 0 ? w : y > 0 ? n : 0;
	p.n = y > 0 ? n : p.w;
	p.nw = x > 0 && y > 0 ? nw : p.w;
	p.ne = x + 1 < cw && y > 0 ? ne : p.n;
	p.ww = x > 1 ? ww : p.w;
	p.nww = x > 1 && y > 0 ? nww : p.ww;
	p.nn = p.n; p.nee = p.ne;   // (no channel left as residuals predicts from them: lf_row_plan_channel)
	const ModWP no_wp = ModWP();
	uint32_t err = 0;
	return res + mod_predict(predictor, no_wp, p, &err);
}
// the first `limit` samples of a channel, in place, one after the other; returns the first position that leaves the int16 range (-1: none)
J40_DEV int32_t lf_predict_channel_serial(J40_GLOBAL int16_t *plane, int32_t cw, int32_t limit, int32_t predictor) {
	int32_t first = -1;
	for (int32_t pos = 0; pos < limit; ++pos) {
		const int32_t y = pos / cw, x = pos - y * cw;
		const J40_GLOBAL int16_t *up = plane + (pos - x - cw), *row = plane + (pos - x);
		const int32_t v = lf_predict_value(plane[pos], predictor, x, y, cw, x > 0 ? row[x - 1] : 0, x > 1 ? row[x - 2] : 0, x > 1 && y > 0 ? up[x - 2] : 0,
			x > 0 && y > 0 ? up[x - 1] : 0, y > 0 ? up[x] : 0, x + 1 < cw && y > 0 ? up[x + 1] : 0);
		if ((v < -32768 || v > 32767) && first < 0) first = pos;
		plane[pos] = (int16_t) v;
	}
	return first;
}
// a section's RAW channels in stream order (serial); returns the section's status: "povf" if a sample before the place the lane
// stopped leaves the range, else the lane's
J40_DEV uint32_t lf_predict_section_serial(const J40_GLOBAL DevLfTask &t, uint32_t status, int32_t nb_varblocks, uint32_t raw_mask, uint32_t stopped_at) {
	for (int32_t chan = 0; chan < 7; ++chan) {
		const int32_t nib = (int32_t) ((raw_mask >> (4 * chan)) & 15u);
		if (nib < 2) continue;   // samples already (0), or nothing to add (predictor 0)
		int32_t cw, chh;
		J40_GLOBAL int16_t *plane = lf_channel_plane(t, chan, nb_varblocks, &cw, &chh);
		if (lf_predict_channel_serial(plane, cw, lf_channel_complete(stopped_at, chan, cw, chh), nib - 1) >= 0) return ERR_POVF;
	}
	return status;
}

// the copy a step asked for, by the lane itself (tests/hostsim; the kernel's lanes do it together: lf_row_flush_wave)
J40_DEV void lf_row_flush_serial(LfRowLane &L) {
	for (int32_t i = 0; i < L.flush_n; ++i) L.flush_dst[i] = L.win[i];
	L.flush_n = 0;
}

} // namespace j40hip
