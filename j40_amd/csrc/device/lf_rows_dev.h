// j40_amd/csrc/device/lf_rows_dev.h -- the LfGroup sections of VarDCT frames, one section per wavefront LANE, with everything a
// sample's dependent chain touches held in LDS (k_lf_rows, lf_decode.hip; SURVEY.md 8f-1: the two Modular sub-images of
// j40__lf_group, j40.h:6722-6790 = j40__modular_channel, j40.h:4127-4240, over the LF coefficient image and the HF metadata image).
//
// lf_lanes_dev.h's decoder (same decomposition: a lane is a section, an iteration is one sample of every lane) spent ~2600 cycles
// per sample on ~130 instructions: what it waited for was memory. Its alias entry came from global memory (an L2 round trip on the
// chain of every sample), the row above was loaded from global memory two samples ahead, every sample was a 2-byte global store,
// and on gfx9 one counter (vmcnt) covers loads AND stores: each wait for the alias entry also waited for the store of the sample
// before. Here no sample of a row touches vector memory:
//   * the alias tables of the lane's frame sit in LDS beside its tree, 16 bytes per bucket (28 KB for seven clusters of 256) in a form
//     made for the straight-line step (round 6, "fast entries" below): both symbols a bucket can yield come with their frequency,
//     their offset and their HYBRID INTEGER worked out -- how many extra bits the token asks for and the value those bits are added
//     to --, so that a sample's symbol is one 16-byte read, two selects and a handful of bit-field moves;
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
	const J40_LDS uint32_t *fast;      // the fast entries, four words per bucket ([cluster << log_alpha | bucket]): the straight-line step's
	const J40_GLOBAL uint64_t *alias;  // the alias tables as the host built them, in global memory: the general step's (one sample in sixty)
	int32_t log_alpha, log_bucket;
	uint32_t uses;                     // LfLaneFrame::uses
};

// A fast entry: what the straight-line step needs of a bucket, for the symbol the bucket keeps ("left": the bucket's own index) and the
// one it is aliased to ("right"), made from the host's 8-byte alias entry and the cluster's hybrid-integer configuration when the
// tables are staged (j40.h:2441-2466 and 2313-2334 evaluated per symbol instead of per sample):
//   word 0: cutoff (8 bits) | left frequency (13) << 8 | left extra bits (5) << 21 | left "iovf" << 26
//   word 1: right frequency (13) | right extra bits (5) << 13 | right "iovf" << 18 | right offset (12) << 19
//   word 2, 3: the left / right token's value with all its extra bits zero -- the token itself below the split; a token above the
//     cluster's max_token ("iovf", an error of the stream) gets 2^30, which no plane holds: LF_FAST_IOVF_BASE
// so that (word 0 >> 8) and word 1 have one layout: frequency, extra bits, "iovf", offset (zero on the left).
// The extra bits come out of the low dword of the bit window together with a renormalisation's 16: a cluster whose tokens can ask for
// more than 16 is LF_LEAF_WIDE and keeps the general step (entries of such a cluster are never read).
typedef uint32_t LfFastQuad __attribute__((vector_size(16)));
enum { LF_FAST_IOVF = 1u << 18, LF_FAST_IOVF_BASE = 1u << 30 };
J40_DEV void lf_rows_token_info(uint32_t tok, uint32_t cfg, uint32_t *base, uint32_t *extra, uint32_t *iovf) {
	const uint32_t split_exp = cfg & 15u, split = 1u << split_exp, msb = (cfg >> 4) & 15u, lsb = (cfg >> 8) & 15u, in_token = msb + lsb, mt = cfg >> 12;
	*base = tok; *extra = 0; *iovf = 0;
	if (tok < split) return;
	if (tok > mt) { *base = (uint32_t) LF_FAST_IOVF_BASE; *iovf = 1; return; }
	const uint32_t midbits = split_exp - in_token + ((tok - split) >> in_token);   // (split_exp >= in_token: j40.h:2313)
	if (midbits > 16u) { *base = 0; return; }   // (a wide cluster's: not read)
	const uint32_t top = 1u << msb, lo = tok & ((1u << lsb) - 1u), hi = (tok >> lsb) & (top - 1u);
	*base = ((top | hi) << (midbits + lsb)) | lo; *extra = midbits;
}
// `e`: the host's entry of `bucket` (lane_symbol_in_cluster reads the same fields), `cfg`: the cluster's configuration word (LaneTables::cluster_cfg)
J40_DEV LfFastQuad lf_rows_fast_entry(uint64_t e, uint32_t bucket, uint32_t cfg) {
	const uint32_t elo = (uint32_t) e, ehi = (uint32_t) (e >> 32);
	const uint32_t cutoff = elo & 0xffu, off_r = (elo >> 8) & 0xfffu, tok_r = (elo >> 20) & 0xffu, d_r = (uint32_t) (e >> 28) & 0x1fffu, d_l = (ehi >> 9) & 0x1fffu;
	uint32_t base_l, extra_l, iovf_l, base_r, extra_r, iovf_r;
	lf_rows_token_info(bucket, cfg, &base_l, &extra_l, &iovf_l);
	lf_rows_token_info(tok_r, cfg, &base_r, &extra_r, &iovf_r);
	LfFastQuad q;
	q[0] = cutoff | (d_l << 8) | (extra_l << 21) | (iovf_l << 26);
	q[1] = d_r | (extra_r << 13) | (iovf_r << 18) | (off_r << 19);
	q[2] = base_l; q[3] = base_r;
	return q;
}

// what a staged leaf carries instead of its context: `cfg` as in LaneTables::cluster_cfg (max_token from bit 12 up; tokens are
// < 256, so clamping it to 11 bits keeps `token > max_token` intact), the cluster in the top byte, and bit 23 when a token of the
// cluster can ask for more than 16 extra bits (with a renormalisation's 16 more than the window's low dword: not for the straight-line step)
enum { LF_LEAF_CFG_MASK = 0x7fffff, LF_LEAF_WIDE = 1 << 23 };
J40_DEV int32_t lf_rows_leaf_word(uint32_t cluster, uint32_t cfg) {
	const uint32_t mt = cfg >> 12, split_exp = cfg & 15u, in_token = ((cfg >> 4) & 15u) + ((cfg >> 8) & 15u);
	const uint32_t top_token = mt > 255u ? 255u : mt, split = 1u << split_exp;
	const uint32_t most_extra = top_token >= split ? split_exp - in_token + ((top_token - split) >> in_token) : 0u;   // (split_exp >= in_token: j40.h:2313)
	return (int32_t) ((cfg & 0xfffu) | ((mt > 0x7ffu ? 0x7ffu : mt) << 12) | (most_extra > 16u ? (uint32_t) LF_LEAF_WIDE : 0u) | (cluster << 24));
}

// LDS bytes of one frame's staged tables (tree nodes, then the fast entries)
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t lf_rows_table_bytes(int32_t num_nodes, int32_t num_clusters, int32_t log_alpha) {
	return ((16u * (uint32_t) num_nodes + 15u) & ~15u) + 16u * ((uint32_t) num_clusters << log_alpha);
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
	bool in_run;                       // the lane's last step started a run of plain samples: when it is over, lf_row_run_end (not the general step) follows
	// what the straight-line steps since the last general step found wrong, looked at by the next general step (lf_row_deferred): an
	// error there makes the lane report ERR_LFFB -- the host decodes the section and names the error
	uint32_t acc_range, acc_iovf;      // OR of (sample + 32768): bits from 16 up say a sample left the plane's range; OR of the symbols' LF_FAST_IOVF bits
	bool deferred_real;                // the lane ended on such a finding and it is an error of the stream (not a residual too wide for the plane)
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
	L.plain_left = 0; L.live = true; L.in_run = false; L.acc_range = L.acc_iovf = 0; L.deferred_real = false;
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

// The straight-line step checks nothing by itself: the bit position only grows, so one look at it at the end of a run tells whether any
// symbol of the run read past the section's end ("shrt"); samples outside the plane's range ("povf", or a residual too wide) and tokens
// above max_token ("iovf") leave a bit in an accumulator. Whatever came first, what followed it in the run is garbage (bounded: a run is
// at most LF_ROW_WIN - 1 samples, the codestream is padded for the words such a run can ask for past its end), so the lane cannot
// name the error: it ends with ERR_LFFB, the frame is decoded again by the host's decoder, which does (streams without errors never get
// here; frames with a damaged LfGroup section pay the single-frame path). Nothing of the run's row counts as complete.
J40_DEV bool lf_row_deferred(LfRowLane &L) {
	const bool overrun = lane_bit_position(L.b) > L.end_bit && L.chan < 7 && !L.setup;
	const bool range = (L.acc_range >> 16) != 0, iovf = (L.acc_iovf & (uint32_t) LF_FAST_IOVF) != 0;
	if (!(overrun | range | iovf)) return false;
	L.deferred_real = overrun | iovf | (range & !L.raw);
	L.x = 0;
	lf_row_fail(L, ERR_LFFB);
	return true;
}

// one sample of the lane's stream (or the start of its next channel). The caller copies a completed piece out (L.flush_n) before
// the lane's next step.
J40_DEV void lf_row_step_general(LfRowLane &L, const J40_GLOBAL DevLfTask &t, const LfRowTables &T) {
	if (lf_row_done(L)) return;
	if (lf_row_deferred(L)) return;   // (the straight-line steps since the last general step ran into something)
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
// lf_row_step_plain_for is the same sample as ONE basic block: the property is a signed sum with per-channel coefficients, the tree
// walk a compare and a select, the refill, the renormalisation, the extra bits, the prediction (a sum, "select" and the clamped
// gradient all computed, one chosen) are selects; errors are looked for when a run is over (lf_row_deferred). A lane takes it when its
// channel has the form lf_row_plan_channel accepts and its first symbol has been read, to the end of its row (lf_row_run_end follows);
// everything else -- channel starts, other trees -- is the general step, which the kernel runs for the lanes that need it after the
// others' plain steps. lf_row_step leaves the number of plain samples that follow (plain_left), so that the choice costs one compare.
// how many samples from here on are plain ones: to the end of the row, in wide rows from the second sample to the end of the window's
// current piece (what follows a run -- the row's or the piece's copy out, the next row's start -- is lf_row_run_end's)
J40_DEV int32_t lf_row_plain_run(const LfRowLane &L) {
	const bool can = L.plain_ok & !L.setup & (L.chan < 7) & (L.state != 0) & (!L.plain_wide | (L.x > 0));
	const int32_t last = L.plain_wide ? mod_min(L.cw - 1, L.x | (LF_ROW_WIN - 1)) : L.cw - 1;   // the run's last sample
	return can ? mod_max(last + 1 - L.x, 0) : 0;
}
// what the general step does behind a row's (or, in wide rows, a window piece's) last sample, for a lane whose run of plain samples
// just got there: the run's findings, the copy out, the next row's start or the channel's end
J40_DEV void lf_row_run_end(LfRowLane &L) {
	if (lf_row_deferred(L)) return;
	const int32_t cw = L.cw, nx = L.x;
	if (nx < cw) {   // (a wide row's piece: nx is a multiple of the window)
		L.flush_n = LF_ROW_WIN; L.flush_dst = L.row + (nx - LF_ROW_WIN);
		return;
	}
	L.flush_n = ((cw - 1) & (LF_ROW_WIN - 1)) + 1; L.flush_dst = L.row + (cw - L.flush_n);
	L.x = 0; L.y += 1; L.row += cw; L.pw = L.pww = 0; L.a0 = L.a1 = 0;
	if (L.y < L.chh) {
		if (cw > LF_ROW_WIN) { L.a2 = L.row[-cw]; L.a3 = L.row[1 - cw]; L.a4 = L.row[2 - cw]; }   // (a wide row's first sample is the general step's: W falls back to N there)
		else { L.a2 = L.win[0]; L.a3 = cw > 1 ? L.win[1] : 0; L.a4 = cw > 2 ? L.win[2] : 0; }
		return;
	}
	++L.chan; L.setup = true;
}
// the step the kernel gives the lanes that are not in a run of plain samples; leaves the length of the run that follows
J40_DEV void lf_row_step(LfRowLane &L, const J40_GLOBAL DevLfTask &t, const LfRowTables &T) {
	if (L.in_run) lf_row_run_end(L); else lf_row_step_general(L, t, T);
	L.plain_left = lf_row_plain_run(L);
	L.in_run = L.plain_left > 0;
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

// Round 5 measured (and kept out of the default, since removed) two sections per lane side by side: 421-450 ms per launch against
// 212-222 -- a lone wavefront is bound by how fast it is fed instructions (six to nine cycles each), not by the dependences between
// them, and only fewer instructions per sample help. So: the step is instantiated for what the lanes of the wavefront need of it,
// taken together (wave-uniform; lf_plain_needs, recomputed whenever a lane has been through the general step) -- a wavefront whose
// lanes all sit in channels left as residuals runs no neighbour, no property and no prediction at all --, and (round 6, second half)
// the symbol itself comes out of a FAST ENTRY (above): the hybrid integer is worked out per symbol when the tables are staged, the
// two bit-window reads of a symbol (a renormalisation's 16 bits, the token's extra bits) are one shift, nothing is checked per sample
// (lf_row_deferred) and the run's length is known before it starts (lf_row_run_plain_for): 105 -> ~60 instructions for a residual sample.
enum { LF_NEED_TEST = 1,    // some lane's channel has a test (two different leaf words)
       LF_NEED_LIN = 2, LF_NEED_SEL = 4, LF_NEED_GRAD = 8,   // the predictions some lane uses: a (halved) sum of neighbours, "select", the clamped gradient
       LF_NEED_MUL = 16,    // some lane's leaves have a multiplier other than 1 or an offset
       LF_NEED_ALL = 31,
       LF_NEED_PRED = LF_NEED_TEST | LF_NEED_LIN | LF_NEED_SEL | LF_NEED_GRAD };   // none of these: every lane of the step is in a RAW channel

#ifdef __HIPCC__
J40_DEV uint32_t lf_bfe(uint32_t v, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(v, off, width); }   // (off + width <= 32)
// (state << 16) | (window & 0xffff) for a state below 2^16, one byte permute
J40_DEV uint32_t lf_renorm_word(uint32_t state, uint32_t window) { return __builtin_amdgcn_perm(state, window, 0x05040100u); }
#else
J40_DEV uint32_t lf_bfe(uint32_t v, uint32_t off, uint32_t width) { return width ? (v >> off) & (0xffffffffu >> (32u - width)) : 0u; }
J40_DEV uint32_t lf_renorm_word(uint32_t state, uint32_t window) { return (state << 16) | (window & 0xffffu); }
#endif

// One sample, one basic block. The lane's channel has the form lf_row_plan_channel accepts, its first symbol has been read and the
// sample is not the last of its row (lf_row_plain_run). Checks nothing (lf_row_deferred does, after the run) and leaves plain_left
// alone (the caller counts). `wp`: the sample's slot of the lane's window (a run never wraps around the window: lf_row_plain_run), moved
// on by the step -- as is L.x when something of the step looks at it; when nothing does (residuals only) the caller adds the run's length.
template <uint32_t NEED>
J40_DEV void lf_row_step_plain_for(LfRowLane &L, const LfRowTables &T, J40_LDS int16_t *&wp) {
	LaneBits &b = L.b;
	{   // lane_bits_refill as selects; the word after next is asked for every time (the same word again when nothing was appended)
		const bool need = b.nbits <= 32;
		const uint32_t inc = need ? 4u : 0u;
		b.bits |= (uint64_t) (need ? b.ahead : 0u) << (b.nbits & 63);   // (nothing appended: zero, shifted by whatever)
		b.nbits += (int32_t) (8u * inc); b.pos += inc;
		b.ahead = lane_load32(b.base, b.pos);
	}
	uint32_t word = L.k_word_le;   // (no lane tests anything: both words are the leaf's)
	int32_t pw = 0, pn = 0, pnw = 0, pne = 0, pww = 0, ahead3 = 0;
	if (NEED & LF_NEED_PRED) {
		const int32_t x = L.x, y = L.y, cw = L.cw;
		L.x = x + 1;
		ahead3 = wp[3];   // the row above at x + 3, for the next sample's registers (slots past the row's end are never looked at)
		const bool up = y > 0, left = x > 0;
		pw = left ? L.pw : up ? L.a2 : 0;
		pn = up ? L.a2 : pw;
		pnw = left && up ? L.a1 : pw;
		pne = x + 1 < cw && up ? L.a3 : pn;
		pww = x > 1 ? L.pww : pw;
		// the property and the walk: one test, or none
		if (NEED & LF_NEED_TEST) {
			const int32_t pnww = x > 1 && up ? L.a0 : pww;
			int32_t val = lf_mad24(L.c_x, x, lf_mad24(L.c_y, y, lf_mad24(L.c_w, pw, lf_mad24(L.c_n, pn, 0)))) + lf_mad24(L.c_nw, pnw, lf_mad24(L.c_ne, pne, lf_mad24(L.c_ww, pww, lf_mad24(L.c_nww, pnww, 0))));   // (two chains of four)
			val = L.c_abs ? mod_abs(val) : val;
			val = L.c_first && !left ? pw : val;
			word = val > L.k_thr ? L.k_word_gt : L.k_word_le;
		}
	}
	// the symbol: rANS step (j40.h:2441-2466) and hybrid integer (j40.h:2313-2334) from the fast entry of the state's bucket in the leaf's cluster
	const uint32_t st = L.state, bucket = lf_bfe(st, (uint32_t) T.log_bucket, (uint32_t) T.log_alpha), pos = st & ((1u << T.log_bucket) - 1u);
	const LfFastQuad e = *(const J40_LDS LfFastQuad *) (T.fast + 4u * ((word >> 24) << T.log_alpha) + 4u * bucket);
	const bool aliased = pos >= (e[0] & 0xffu);
	const uint32_t w = aliased ? e[1] : e[0] >> 8, base = aliased ? e[3] : e[2];
	const uint32_t state = (w & 0x1fffu) * (st >> 12) + (w >> 19) + pos;   // (both factors below 2^24; bits 19 up: the offset, nothing above it)
	const bool renorm = state < (1u << 16);
	const uint32_t window = (uint32_t) b.bits, skip = renorm ? 16u : 0u, extra = (w >> 13) & 31u;   // (<= 16 + 16 of the > 32 bits the window holds)
	L.state = renorm ? lf_renorm_word(state, window) : state;
	const uint32_t mid = lf_bfe(window, skip, extra), taken = skip + extra;
	b.bits >>= taken; b.nbits -= (int32_t) taken;
	const int32_t u = (int32_t) (base + (mid << ((word >> 8) & 15u)));   // (the token's value: its bits are apart from the extra bits')
	// the prediction
	int32_t pred = 0;
	if (NEED & LF_NEED_LIN) {
		int32_t lin = lf_mad24(L.p_w, pw, lf_mad24(L.p_n, pn, 0)) + lf_mad24(L.p_nw, pnw, lf_mad24(L.p_ne, pne, lf_mad24(L.p_ww, pww, 0)));
		pred = L.p_half ? (lin + (int32_t) ((uint32_t) lin >> 31)) >> 1 : lin;   // (a + b) / 2, towards zero
	}
	if (NEED & LF_NEED_SEL) {
		const int32_t sel = mod_abs(pn - pnw) < mod_abs(pw - pnw) ? pw : pn;
		pred = (NEED & (LF_NEED_LIN | LF_NEED_GRAD)) ? (L.p_kind == 1 ? sel : pred) : sel;   // (the only kind around: no lane to tell apart)
	}
	if (NEED & LF_NEED_GRAD) {
		const int32_t grad = mod_gradient(pw, pn, pnw);
		pred = (NEED & (LF_NEED_LIN | LF_NEED_SEL)) ? (L.p_kind == 2 ? grad : pred) : grad;
	}
	const int32_t res = (NEED & LF_NEED_MUL) ? lf_unzigzag(u) * L.p_mul + L.p_off : lf_unzigzag(u);
	const int32_t v = (NEED & LF_NEED_PRED) ? res + (L.raw ? 0 : pred) : res;
	// what lf_row_deferred looks at: the sample's range (LF_FAST_IOVF_BASE is outside it unless a multiplier folds it back: then the bit)
	L.acc_range |= (uint32_t) (v + 32768);
	if (NEED & LF_NEED_MUL) L.acc_iovf |= w;
	*wp++ = (int16_t) v;
	if (NEED & LF_NEED_PRED) {
		L.pww = L.pw; L.pw = v;
		L.a0 = L.a1; L.a1 = L.a2; L.a2 = L.a3; L.a3 = L.a4; L.a4 = ahead3;
	}
}

// this lane's share of the wavefront's needs (0 for a lane that takes no plain step)
J40_DEV uint32_t lf_plain_needs(const LfRowLane &L) {
	if (!(L.live & L.plain_ok)) return 0;
	if (L.raw) return L.p_mul != 1 || L.p_off != 0 ? (uint32_t) LF_NEED_MUL : 0u;   // (nothing of the prediction)
	return (L.k_word_gt != L.k_word_le ? (uint32_t) LF_NEED_TEST : 0u) | (L.p_kind == 0 ? (uint32_t) LF_NEED_LIN : L.p_kind == 1 ? (uint32_t) LF_NEED_SEL : (uint32_t) LF_NEED_GRAD)
		| (L.p_mul != 1 || L.p_off != 0 ? (uint32_t) LF_NEED_MUL : 0u);
}
// one sample of a lane inside its run, through the step for lanes whose needs add up to `need` (tests/hostsim; the kernel: lf_row_run_plain)
#define LF_PLAIN_CASE(n) case n: lf_row_step_plain_for<n>(L, T, wp); break;
J40_DEV void lf_row_step_plain_needs(LfRowLane &L, const LfRowTables &T, uint32_t need) {
	J40_LDS int16_t *wp = L.win + (L.x & (LF_ROW_WIN - 1));
	if (!(need & (uint32_t) LF_NEED_PRED)) L.x += 1;
	switch (need) {
	LF_PLAIN_CASE(0) LF_PLAIN_CASE(1) LF_PLAIN_CASE(2) LF_PLAIN_CASE(3) LF_PLAIN_CASE(4) LF_PLAIN_CASE(5) LF_PLAIN_CASE(6) LF_PLAIN_CASE(7)
	LF_PLAIN_CASE(8) LF_PLAIN_CASE(9) LF_PLAIN_CASE(10) LF_PLAIN_CASE(11) LF_PLAIN_CASE(12) LF_PLAIN_CASE(13) LF_PLAIN_CASE(14) LF_PLAIN_CASE(15)
	LF_PLAIN_CASE(16) LF_PLAIN_CASE(17) LF_PLAIN_CASE(18) LF_PLAIN_CASE(19) LF_PLAIN_CASE(20) LF_PLAIN_CASE(21) LF_PLAIN_CASE(22) LF_PLAIN_CASE(23)
	LF_PLAIN_CASE(24) LF_PLAIN_CASE(25) LF_PLAIN_CASE(26) LF_PLAIN_CASE(27) LF_PLAIN_CASE(28) LF_PLAIN_CASE(29) LF_PLAIN_CASE(30)
	default: lf_row_step_plain_for<LF_NEED_ALL>(L, T, wp); break;
	}
	L.plain_left -= 1;
}
#undef LF_PLAIN_CASE
#ifdef __HIPCC__
// The kernel's inner loop: the wavefront's lanes step through their runs of plain samples until some live lane is out of its run (a
// channel start, a row's end: the caller's business). Nothing inside a run can end it early (lf_row_deferred), so its length is known
// beforehand: the shortest run among the lanes that are in one -- or ONE step when some live lane is not (a lane that stays in a
// channel of another form does not hold the others up, and they do not hold it up either). The loop is a scalar count; the choice among
// the instantiations is made once per such stretch (a wave-uniform switch is a tree of scalar branches, too dear to walk per sample).
template <uint32_t NEED>
J40_DEV void lf_row_run_plain_for(LfRowLane &L, const LfRowTables &T) {
	const bool plain = L.plain_left > 0;
	if (!__builtin_amdgcn_ballot_w64(plain)) return;
	int32_t shortest = plain ? L.plain_left : 0x7fffffff;
	for (int32_t d = 32; d >= 1; d >>= 1) shortest = mod_min(shortest, __shfl_xor(shortest, d));
	const int32_t n = __builtin_amdgcn_ballot_w64(!plain & L.live) ? 1 : __builtin_amdgcn_readfirstlane(shortest);
	if (plain) {
		J40_LDS int16_t *wp = L.win + (L.x & (LF_ROW_WIN - 1));   // (= x unless the row is wider than the window)
		for (int32_t i = 0; i < n; ++i) lf_row_step_plain_for<NEED>(L, T, wp);
		if (!(NEED & LF_NEED_PRED)) L.x += n;
		L.plain_left -= n;
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
#undef LF_PLAIN_CASE
#endif

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
	if (status == (uint32_t) ERR_LFFB) return status;   // (the lane gave up, perhaps over garbage: the host decodes the section)
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
