// j40_amd/csrc/device/hf_lanes_dev.h -- K1, throughput form: the HF coefficient decode of one pass-group
// section per wavefront LANE, specialised for what real VarDCT streams use (rANS, no LZ77). Restates
// j40__hf_coeffs (j40.h:6888-7005) + j40__code / j40__ans_code / j40__hybrid_int (2804, 2441, 2313) as a
// flat state machine: every iteration each of the 64 lanes decodes exactly one symbol -- the non-zero
// count of its next (block, channel) or the next coefficient -- so the rANS step, the hybrid integer and
// the bit refill are executed convergently; only the short per-phase prologue / epilogue diverge.
//
// What makes an iteration cheap:
//   * branch-light: a divergent `if` costs exec-mask bookkeeping plus a branch whether or not it is taken,
//     so the per-symbol path is written with selects. One refill point per iteration tops the 64-bit window
//     up to > 32 bits without a branch; the rANS renormalisation and the hybrid integer's extra bits are
//     taken unconditionally with a length of 0 when they do not apply. Only the rare cases branch: first
//     symbol of a section, more than ~17 extra bits, errors.
//   * the bit position is absolute (8 * pos - nbits), so "ran past the section" is one compare per symbol
//     instead of end-of-section logic inside every read; the padded codestream makes over-reads harmless;
//   * per symbol: context -> cluster (LDS byte), then the 64-bit alias entry and the cluster's hybrid-integer
//     config word side by side (both LDS, both addressed by the cluster): a chain of two LDS reads;
//   * every pointer carries its address space (LDS tables, global block lists / bitstream / coefficients) --
//     no flat accesses; per-lane addresses are 32-bit offsets from wave-uniform bases;
//   * the non-zero-count predictor (left / top neighbour, j40.h:6959-6963) needs only the count of the last block
//     written into each of the group's 32 cell columns: blocks arrive in raster order of their top-left cell and
//     tile the group, so "last block in column x - 1" covers (x - 1, y) and "last block in column x" covers
//     (x, y - 1). 96 bytes per lane in LDS replace the 3 KB per group scratch in HBM and its dependent loads;
//   * the next block's descriptor is requested while the current block is being decoded; the block contexts
//     come with the descriptor (looked up on the host);
//   * the three coefficient planes are one allocation (plane c at c * coeff_stride), so a lane's channel is
//     an offset, not a pointer select.
// Specs with prefix codes or LZ77, and tables beyond the LDS budget, take decode_hf_section_flat (hf_dev.h).
#pragma once
#include "hf_dev.h"

namespace j40hip {

// LDS-resident tables of one frame / pass (staged by the kernel, or plain arrays under tests/hostsim)
struct LaneTables {
	const J40_LDS uint8_t *ctx_map;       // [num_dist] context -> cluster
	const J40_LDS uint32_t *cluster_cfg;  // [cluster] split_exp | msb_in_token << 4 | lsb_in_token << 8 | max_token << 12
	const J40_LDS uint64_t *alias;        // [cluster << log_alpha_size | bucket], AnsEntry (entropy.hpp)
	const J40_LDS int16_t *nnz_ctx2;
	const J40_LDS int8_t *freq_ctx2;
	const J40_LDS uint32_t *dct_info;     // [DctSelect] log_rows | log_columns << 8 | order_idx << 16
	int32_t log_alpha, log_bucket;
};

// the frame scalars the decoder needs, copied out of DevFrame once (wave-uniform)
struct LaneFrame {
	int32_t nb_block_ctx, num_hf_presets, preset_bits, check_section_end;   // (DevFrame::check_section_end)
	uint32_t single_declared_end;
	const J40_GLOBAL uint32_t *order_off;  // DevFrame::order_off
};

// wave-uniform global bases of one frame; per-lane state only holds 32-bit offsets into them
struct LaneGlobals {
	const J40_GLOBAL uint8_t *codestream;
	const J40_GLOBAL uint32_t *group_blocks;   // DevGroupBlock as two words
	J40_GLOBAL float *coeffs;                  // dense planes (multi-pass frames): plane c at c * coeff_stride
	J40_GLOBAL CoeffEvent *events;             // sparse coefficients (single-pass frames), see DevPlan::events
	J40_GLOBAL uint32_t *block_events;
	const J40_GLOBAL uint16_t *pool_u16;       // coefficient orders (multi-pass frames)
	uint32_t coeff_stride;
};

// LSB-first bit window over the 4-byte aligned codestream buffer. Absolute position of the next unread
// bit = 8 * pos - nbits. Reads never fail; the caller compares the position with the section end.
struct LaneBits {
	const J40_GLOBAL uint8_t *base;
	uint64_t bits;     // bits at and above nbits are zero
	int32_t nbits;
	uint32_t pos;      // byte offset of the next word to append (multiple of 4)
	uint32_t ahead;    // the word at pos, requested one refill early
};

J40_DEV uint32_t lane_load32(const J40_GLOBAL uint8_t *base, uint32_t pos) { return *(const J40_GLOBAL uint32_t *) (base + pos); }

J40_DEV void lane_bits_init(LaneBits &b, const J40_GLOBAL uint8_t *base, uint32_t start_bit) {
	b.base = base;
	const uint32_t pos0 = (start_bit >> 3) & ~3u, skip = start_bit - 8u * pos0;   // skip < 32
	b.bits = (uint64_t) (lane_load32(base, pos0) >> skip);
	b.nbits = 32 - (int32_t) skip;
	b.pos = pos0 + 4;
	b.ahead = lane_load32(base, b.pos);
}

// > 32 bits buffered afterwards. The append is needed once in several symbols; it is a (divergent) branch so that the load of
// the word after next is issued only then and has all the symbols until the next append to arrive: requesting it in every
// iteration made every iteration wait for the request of the one before (and, vmcnt being shared, for its stores)
J40_DEV void lane_bits_refill(LaneBits &b) {
	if (b.nbits <= 32) {
		b.bits |= (uint64_t) b.ahead << b.nbits;
		b.nbits += 32;
		b.pos += 4u;
		b.ahead = lane_load32(b.base, b.pos);
	}
}

J40_DEV uint32_t lane_bits_take(LaneBits &b, int32_t n) {   // 0 <= n <= 31, n <= nbits
	const uint32_t v = (uint32_t) b.bits & ((1u << n) - 1u);
	b.bits >>= n; b.nbits -= n;
	return v;
}

J40_DEV uint32_t lane_bit_position(const LaneBits &b) { return 8u * b.pos - (uint32_t) b.nbits; }

// one symbol: rANS step (j40.h:2441-2466) + hybrid integer (j40.h:2313-2334). *err receives the error the
// reference would have raised first ("shrt" while renormalising, then "iovf", then "shrt" in the extra bits).
// `alias`: the tables of all clusters ([cluster << log_alpha | bucket]); cl, m: the symbol's cluster and that cluster's
// configuration word (LaneTables::cluster_cfg) -- looked up by the caller, who may have them at hand already (lf_rows_dev.h)
// STRAIGHT: the caller vouches that the state has been read (state != 0) and that the cluster's tokens never ask for more than 17
// extra bits (33 buffered - 16 of a renormalisation): the symbol is then one basic block, no branch at all -- a lone wavefront pays
// about sixty cycles for every `if` it walks past, taken or not (lf_rows_dev.h)
// STARTED: only the first of the two is vouched for (the state has been read; a token may still ask for a second refill).
template <bool STRAIGHT = false, bool STARTED = false, class AliasPtr>
J40_DEV int32_t lane_symbol_in_cluster(LaneBits &b, uint32_t &state, AliasPtr alias, int32_t log_alpha, int32_t log_bucket, uint32_t cl, uint32_t m, uint32_t end_bit, uint32_t *err) {
	if (!STRAIGHT && !STARTED && state == 0) {   // first symbol of the section (j40.h:2445-2449); the window holds > 32 bits
		state = lane_bits_take(b, 16); state |= lane_bits_take(b, 16) << 16;
		lane_bits_refill(b);
	}
	const uint32_t idx = state & 0xfff, i = idx >> log_bucket, pos = idx & ((1u << log_bucket) - 1);
	const uint64_t e = alias[(cl << log_alpha) + i];
	const uint32_t elo = (uint32_t) e, ehi = (uint32_t) (e >> 32);
	const bool aliased = pos >= (elo & 0xff);
	const int32_t token = (int32_t) (aliased ? (elo >> 20) & 0xff : i);
	const uint32_t offset = aliased ? (elo >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (ehi >> 9) & 0x1fff;
	state = d * (state >> 12) + offset + pos;
	const bool renorm = state < (1u << 16);
	const uint32_t low = lane_bits_take(b, renorm ? 16 : 0);
	state = renorm ? (state << 16) | low : state;
	const bool short1 = lane_bit_position(b) > end_bit;
	// hybrid integer, computed for every token and selected at the end
	const int32_t split_exp = (int32_t) (m & 15), split = 1 << split_exp;
	const bool big = token >= split;
	const int32_t mt = (int32_t) (m >> 12);
	const bool iovf = big && token > mt;
	const int32_t tok = iovf ? mt : token;
	const int32_t msb = (int32_t) ((m >> 4) & 15), lsb = (int32_t) ((m >> 8) & 15), in_token = msb + lsb;
	const int32_t midbits = big ? split_exp - in_token + ((tok - split) >> in_token) : 0;
	if (!STRAIGHT && midbits > b.nbits) lane_bits_refill(b);   // rare: more than ~17 extra bits (nbits <= 31 here, so the refill appends)
	const int32_t mid = (int32_t) lane_bits_take(b, midbits);
	const bool short2 = lane_bit_position(b) > end_bit;
	const int32_t top = 1 << msb;
	const int32_t lo = tok & ((1 << lsb) - 1), hi = (tok >> lsb) & (top - 1);
	const int32_t value = ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
	*err = short1 ? (uint32_t) ERR_SHRT : iovf ? (uint32_t) ERR_IOVF : short2 ? (uint32_t) ERR_SHRT : 0u;
	return big ? value : token;
}
template <class Tables>   // LaneTables, or a set of tables with the same members elsewhere (LfLaneTables, lf_lanes_dev.h)
J40_DEV int32_t lane_symbol(LaneBits &b, uint32_t &state, const Tables &t, int32_t ctx, uint32_t end_bit, uint32_t *err) {
	const uint32_t cl = t.ctx_map[ctx];
	return lane_symbol_in_cluster(b, state, t.alias, t.log_alpha, t.log_bucket, cl, t.cluster_cfg[cl], end_bit, err);
}

typedef uint32_t LaneEventQuad __attribute__((vector_size(16)));   // four events, one 16-byte store

// the entry of DevPlan::block_events of a finished block {first event, n_Y, n_X, n_B}, one 16-byte store
J40_DEV void lane_store_block_events(const LaneGlobals &G, uint32_t blk, uint32_t first, uint32_t n0, uint32_t n1, uint32_t n2) {
	J40_GLOBAL uint64_t *e = (J40_GLOBAL uint64_t *) (G.block_events + 4u * blk);
	e[0] = (uint64_t) first | ((uint64_t) n0 << 32); e[1] = (uint64_t) n1 | ((uint64_t) n2 << 32);
}

// what the decoder needs to start on a section
struct LaneSection {
	uint32_t start_bit, end_bit;       // the section's bits in the codestream
	uint32_t cell_base;                // of its LfGroup
	uint32_t block_first; int32_t nblocks;   // its group's block list
	uint32_t ev_first, ev_end;         // its region of the event list (sparse coefficients)
};

// Decodes the sections `src` hands out, one after the other, on this lane; each section's status goes back through src.done.
// `Source`: bool next(LaneSection &) -- false: nothing left for this lane --, void done(uint32_t status, uint32_t end_bit).
// A lane that finishes a section takes the next one INSIDE the symbol loop, so the other lanes of its wavefront never wait for it:
// with more sections than lanes (sections handed out by decreasing size through a counter the lanes of a frame share, kernels.hip:
// k_hf_lanes) a wavefront's lanes stay busy until the frame runs out of sections, instead of idling behind its longest one.
// Same results and status codes per section as decode_hf_section (hf_dev.h).
// cols[(c * 32 + x) * col_stride]: non-zero count (per 8x8 cell) of the last block written into cell column x, channel c. It needs
// no reset between sections: a block reads only columns a block of its own section wrote before it (blocks tile the group in
// raster order of their top-left cells).
// ring[(n % HF_LANE_RING_SLOTS) * ring_stride]: the lane's n-th event until it has left for global memory (SCAN). Events are written
// there and leave J40_LANE_EV_FLUSH at a time as ONE aligned store (the regions of DevPlan::events start at multiples of 32 events):
// a lane's 4-byte stores, each into a line of its own that its next store reaches hundreds of cycles later, left the L2 as partly
// written sectors over and over -- 21 GB of writes per 256 8K frames for 4 GB of events. The check runs every J40_LANE_EV_FLUSH-th
// turn (a turn adds at most one event, so twice that many slots suffice); what is left at a section's end leaves word by word.
#ifndef J40_LANE_NZ_PERIOD
#define J40_LANE_NZ_PERIOD 8
#endif
#ifndef J40_LANE_STRAIGHT_COEFFS
#define J40_LANE_STRAIGHT_COEFFS 1
#endif
#ifndef J40_LANE_REFILL_SELECTS
#define J40_LANE_REFILL_SELECTS 0   // (measured with the event ring, whose turns store nothing: call K)
#endif
template <bool SCAN, class Source>
J40_DEV void decode_hf_sections_lane(const LaneFrame &f, const LaneTables &t, const LaneGlobals &G, Source &src, J40_LDS int8_t *cols, int32_t col_stride, int32_t pass, J40_LDS uint32_t *ring = nullptr, int32_t ring_stride = 0) {
	LaneSection S;
	S.start_bit = S.end_bit = S.cell_base = S.block_first = S.ev_first = S.ev_end = 0; S.nblocks = 0;
	LaneBits b;
	b.base = G.codestream; b.bits = 0; b.nbits = 0; b.pos = 0; b.ahead = 0;
	const int32_t nb_block_ctx = f.nb_block_ctx;
	uint32_t err = 0, end_bit = 0, cell64 = 0, block_first = 0;
	int32_t nblocks = 0, ctxoff = 0;
	bool have = false;   // a section is being decoded
	int32_t k = 0, c_yxb = 0;
	bool in_coeffs = false, done = true;
	uint32_t state = 0;
	int32_t x8 = 0, y8 = 0, log_columns = 3, order_idx = 0, shift = 0, size = 64;
	uint32_t coeffoff = 0, coeff_at = 0, bctx3 = 0;
	int32_t c = 1, bctx = 0, nz = 0, i = 0, prev = 0, cctx = 0;
	const J40_GLOBAL uint16_t *order = nullptr;
	uint32_t ev_at = 0, ev_end = 0, chan_first = 0;   // SCAN: next free event of this section's region, first event of the current channel
	uint32_t ev_flushed = 0;                          // SCAN: events before this one are in global memory
	uint32_t blk_first = 0, blk_n0 = 0, blk_n1 = 0;   // the block's table entry, stored in one piece after its third channel
	uint32_t next0 = 0, next1 = 0;   // descriptor of block k, requested one block ahead
	for (uint32_t turn = 0; ; ++turn) {
#if J40_LANE_EV_FLUSH
		// (the turn is the same in every lane: a scalar branch that skips the per-lane test seven turns in eight -- kept apart from it by
		// the empty statement, or the compiler folds both into one vector condition evaluated every turn)
		if (SCAN && (turn % J40_LANE_EV_FLUSH) == 0) {
#ifdef __HIPCC__
			asm volatile("");
#endif
			if (ev_at - ev_flushed >= (uint32_t) J40_LANE_EV_FLUSH) {
			const J40_LDS uint32_t *r = ring + (int32_t) (ev_flushed % (uint32_t) HF_LANE_RING_SLOTS) * ring_stride;   // (a run of slots: regions and pieces are aligned)
			J40_GLOBAL uint32_t *dst = (J40_GLOBAL uint32_t *) G.events + ev_flushed;
#pragma unroll
			for (int32_t k4 = 0; k4 < J40_LANE_EV_FLUSH; k4 += 4) {
				LaneEventQuad q4;
				q4[0] = r[k4 * ring_stride]; q4[1] = r[(k4 + 1) * ring_stride]; q4[2] = r[(k4 + 2) * ring_stride]; q4[3] = r[(k4 + 3) * ring_stride];
				*(J40_GLOBAL LaneEventQuad *) (dst + k4) = q4;
			}
			ev_flushed += (uint32_t) J40_LANE_EV_FLUSH;
			}
		}
#endif
		// ---- a coefficient of the current (block, channel), sparse form: the symbol nine turns in ten, on a path of its own ----
		// A wavefront alone on its SIMD pays about sixty cycles for every `if` it walks past, executed or not (lf_rows_dev.h), and the
		// turn below -- written for both kinds of symbol, section starts and ends -- is ten of them. A lane inside a channel's
		// coefficients takes this block instead: the refill, the two context look-ups, the symbol (its state has been read: no first-
		// symbol test), the event, the counts; the channel's end, an error or the section's end leave `in_coeffs` / `done` for the
		// general turn, which then runs on every NZ_PERIOD-th turn only and only for the lanes that are not here.
		// (With the event ring: a lane adds at most one event a turn here too -- one that leaves its coefficients in this block takes a
		// count symbol in the general turn, not a coefficient; the one exception would be a state of exactly zero behind a symbol, which
		// sends the lane's next coefficient through the general turn in the same turn: the ring's 2 F slots then hold F - 1 + F + 1.)
		if (SCAN && J40_LANE_STRAIGHT_COEFFS) {
			if (in_coeffs && !done && state != 0) {   // (a state of zero is read afresh, j40.h:2445: the general turn's business, if a stream ever gets there)
				if (J40_LANE_REFILL_SELECTS) {   // lane_bits_refill as selects (lf_rows_dev.h); the word after next is asked for every turn
					const bool need = b.nbits <= 32;
					b.bits |= (uint64_t) (need ? b.ahead : 0u) << (need ? b.nbits : 0);
					b.nbits += need ? 32 : 0; b.pos += need ? 4u : 0u;
					b.ahead = lane_load32(b.base, b.pos);
				} else lane_bits_refill(b);
				const int32_t cctx_now = cctx + t.nnz_ctx2[(nz + (1 << shift) - 1) >> shift] + t.freq_ctx2[i >> shift] + prev;
				uint32_t e2;
				const uint32_t cl = t.ctx_map[cctx_now];
				const int32_t v = lane_symbol_in_cluster<false, true>(b, state, t.alias, t.log_alpha, t.log_bucket, cl, t.cluster_cfg[cl], end_bit, &e2);
				const bool nonzero = v != 0 && e2 == 0;
				const int32_t sv = unpack_signed_dev(v);
				const bool full = nonzero && (ev_at >= ev_end || !coeff_event_fits(sv));
				if (nonzero && !full) {
					if (J40_LANE_EV_FLUSH) ring[(int32_t) (ev_at % (uint32_t) HF_LANE_RING_SLOTS) * ring_stride] = coeff_event_pack((uint32_t) i, sv);
					else ((J40_GLOBAL uint32_t *) G.events)[ev_at] = coeff_event_pack((uint32_t) i, sv);
				}
				ev_at += nonzero && !full ? 1u : 0u;
				e2 = e2 ? e2 : full ? (uint32_t) ERR_EVOF : 0u;
				prev = v != 0;
				nz -= prev;
				++i;
				in_coeffs = nz != 0;
				e2 = e2 ? e2 : in_coeffs && i >= size ? (uint32_t) ERR_COEF : 0u;   // non-zeros left but no coefficient left (j40.h:6996)
				const bool next_channel = !in_coeffs && e2 == 0;
				c_yxb += next_channel ? 1 : 0;
				const bool next_block = next_channel && c_yxb == 3;
				c_yxb = next_block ? 0 : c_yxb;
				k += next_block ? 1 : 0;
				err = e2 ? e2 : err;
				done = e2 != 0 || (next_block && k >= nblocks);
			}
			// (the turn is the same in every lane: a scalar branch seven turns in eight, kept apart from the per-lane test by the empty
			// statement -- folded into one vector condition, every turn would pay for an `if`)
			if ((turn % J40_LANE_NZ_PERIOD) != 0) continue;
#ifdef __HIPCC__
			asm volatile("");
#endif
			if (in_coeffs && !done && state != 0) continue;
		}
		if (done) {   // (rare: twice per section)
			if (have) {
				// ---- the section's end ----
				if (SCAN && J40_LANE_EV_FLUSH) for (; ev_flushed < ev_at; ++ev_flushed) ((J40_GLOBAL uint32_t *) G.events)[ev_flushed] = ring[(int32_t) (ev_flushed % (uint32_t) HF_LANE_RING_SLOTS) * ring_stride];
				if (SCAN && !err && nblocks > 0) lane_store_block_events(G, block_first + (uint32_t) nblocks - 1u, blk_first, blk_n0, blk_n1, ev_at - chan_first);   // the last block
				if (!err) {   // j40.h:2884-2893: the final state, or the untouched initial state, must be 0x130000
					if (state == 0) { lane_bits_refill(b); state = lane_bits_take(b, 16); state |= lane_bits_take(b, 16) << 16; if (lane_bit_position(b) > end_bit) err = ERR_SHRT; }
					if (!err && state != 0x130000) err = ERR_ANS;
				}
				if (!err && f.check_section_end) {   // single-section frames: zero padding up to the byte boundary, then no byte of the section left (j40.h:8203, 7796)
					const uint32_t at = lane_bit_position(b), padn = (0u - at) & 7u;
					if (padn > (uint32_t) b.nbits) lane_bits_refill(b);
					if (lane_bits_take(b, (int32_t) padn)) err = ERR_PAD0;
					else if (at + padn != 8u * f.single_declared_end) err = at + padn < 8u * f.single_declared_end ? (uint32_t) ERR_SHRT : (uint32_t) ERR_EXCS;
				}
				src.done(err, lane_bit_position(b));   // (the position: frames with extra channels, their Modular sub-image starts there -- validate_trailers, runtime.hip)
			}
			have = src.next(S);
			if (!have) break;
			// ---- the section's start ----
			end_bit = S.end_bit; cell64 = S.cell_base * 64u; block_first = S.block_first; nblocks = S.nblocks;
			lane_bits_init(b, G.codestream, S.start_bit);
			lane_bits_refill(b);
			err = 0;
			const uint32_t preset = lane_bits_take(b, f.preset_bits);
			if (lane_bit_position(b) > end_bit) err = ERR_SHRT;
			else if ((int32_t) preset >= f.num_hf_presets) err = ERR_RNGE;
			ctxoff = 495 * nb_block_ctx * (int32_t) preset;
			k = 0; c_yxb = 0; in_coeffs = false; state = 0;
			ev_at = chan_first = blk_first = ev_flushed = S.ev_first; ev_end = S.ev_end; blk_n0 = blk_n1 = 0;
			done = nblocks == 0 || err != 0;
			if (done) continue;   // (nothing to decode: its end is the next turn's business)
			{ const J40_GLOBAL uint32_t *p = G.group_blocks + 2u * block_first; next0 = p[0]; next1 = p[1]; }
		}
		// Block starts are rare (3 per block against dozens of coefficient symbols) but with 64 lanes some lane starts a block in
		// nearly every iteration, and then the whole wavefront walks the block-start code. It is therefore only executed every
		// NZ_PERIOD-th iteration: a lane that reaches a block start in between sits out until then (J40_LANE_NZ_PERIOD = 1: never)
		if (!in_coeffs && (turn % J40_LANE_NZ_PERIOD) != 0) continue;
		lane_bits_refill(b);
		int32_t ctx;
		if (!in_coeffs) {  // next symbol: number of non-zeros of (block k, channel c_yxb), j40.h:6959-6967
			if (SCAN) {   // the channel before this one is complete: book its event count here, off the per-coefficient path
				const uint32_t n = ev_at - chan_first;
				if (c_yxb == 1) blk_n0 = n; else if (c_yxb == 2) blk_n1 = n;
				else if (k > 0) lane_store_block_events(G, block_first + (uint32_t) k - 1u, blk_first, blk_n0, blk_n1, n);
				chan_first = ev_at;
			}
			if (c_yxb == 0) {
				const uint32_t w = next1;
				coeffoff = next0 & ~15u;
				blk_first = ev_at;
				if (k + 1 < nblocks) { const J40_GLOBAL uint32_t *p = G.group_blocks + 2u * (block_first + (uint32_t) k + 1u); next0 = p[0]; next1 = p[1]; }
				x8 = (int32_t) (w & 31); y8 = (int32_t) ((w >> 5) & 31); bctx3 = w >> 16;
				const uint32_t di = t.dct_info[(w >> 10) & 31];
				log_columns = (int32_t) ((di >> 8) & 255); order_idx = (int32_t) (di >> 16);
				shift = (int32_t) (di & 255) + log_columns - 6; size = 64 << shift;
			}
			c = c_yxb == 0 ? 1 : c_yxb == 1 ? 0 : 2;
			bctx = (int32_t) ((bctx3 >> (4 * c_yxb)) & 15);
			const J40_LDS int8_t *col = cols + (c * 32 + x8) * col_stride;
			const int32_t left = x8 > 0 ? col[-col_stride] : -1, topv = y8 > 0 ? col[0] : -1;
			const int32_t pnz = left < 0 ? (topv < 0 ? 32 : topv) : topv < 0 ? left : (left + topv + 1) >> 1;
			ctx = ctxoff + bctx + (pnz < 8 ? pnz : 4 + pnz / 2) * nb_block_ctx;
		} else {
			ctx = cctx + t.nnz_ctx2[(nz + (1 << shift) - 1) >> shift] + t.freq_ctx2[i >> shift] + prev;
		}
		uint32_t e2;   // the first error of this iteration; set instead of leaving the loop midway so that the epilogues stay select-shaped
		const int32_t v = lane_symbol(b, state, t, ctx, end_bit, &e2);
		if (!in_coeffs) {
			nz = v;
			e2 = e2 ? e2 : nz > (63 << shift) ? (uint32_t) ERR_COEF : 0u;
			const int8_t qnz = (int8_t) ((nz + (1 << shift) - 1) >> shift);
			J40_LDS int8_t *col = cols + (c * 32 + x8) * col_stride;
			for (int32_t q = 0; q < (1 << (log_columns - 3)); ++q) col[q * col_stride] = qnz;
			cctx = ctxoff + 458 * bctx + 37 * nb_block_ctx;
			prev = nz <= (size >> 4);
			i = 1 << shift;
			if (!SCAN) {
				coeff_at = (uint32_t) c * G.coeff_stride + cell64 + coeffoff;
				order = G.pool_u16 + f.order_off[(pass * 13 + order_idx) * 3 + c];
			}
			in_coeffs = nz > 0;
		} else {
			const bool nonzero = v != 0 && e2 == 0;
			if (SCAN) {   // one sequential 4-byte store per non-zero coefficient (CoeffEvent: position | value << 16)
				const int32_t sv = unpack_signed_dev(v);
				const bool full = nonzero && (ev_at >= ev_end || !coeff_event_fits(sv));
				if (nonzero && !full) {
					if (J40_LANE_EV_FLUSH) ring[(int32_t) (ev_at % (uint32_t) HF_LANE_RING_SLOTS) * ring_stride] = coeff_event_pack((uint32_t) i, sv);
					else ((J40_GLOBAL uint32_t *) G.events)[ev_at] = coeff_event_pack((uint32_t) i, sv);
				}
				ev_at += nonzero && !full ? 1u : 0u;
				e2 = e2 ? e2 : full ? (uint32_t) ERR_EVOF : 0u;
			} else if (nonzero) G.coeffs[coeff_at + order[i]] += (float) unpack_signed_dev(v);
			prev = v != 0;
			nz -= prev;
			++i;
			in_coeffs = nz != 0;
			e2 = e2 ? e2 : in_coeffs && i >= size ? (uint32_t) ERR_COEF : 0u;   // non-zeros left but no coefficient left (j40.h:6996)
		}
		const bool next_channel = !in_coeffs && e2 == 0;
		c_yxb += next_channel ? 1 : 0;
		const bool next_block = next_channel && c_yxb == 3;
		c_yxb = next_block ? 0 : c_yxb;
		k += next_block ? 1 : 0;
		err = e2 ? e2 : err;
		done = e2 != 0 || (next_block && k >= nblocks);
	}
}

// one section, handed over by the caller (the CPU build of tests/hostsim; frames of one wavefront's worth)
struct LaneOneSection {
	LaneSection S; bool taken; uint32_t status, end_bit;
	J40_DEVM bool next(LaneSection &out) { if (taken) return false; taken = true; out = S; return true; }
	J40_DEVM void done(uint32_t st, uint32_t eb) { status = st; end_bit = eb; }
};

// decodes one (pass, group) section; same results and status codes as decode_hf_section (hf_dev.h).
template <bool SCAN>
J40_DEV uint32_t decode_hf_section_lane(const LaneFrame &f, const LaneTables &t, const LaneGlobals &G, const DevSection &sec, uint32_t cell_base,
		uint32_t block_first, int32_t nblocks, uint32_t ev_first, uint32_t ev_end, J40_LDS int8_t *cols, int32_t col_stride, int32_t pass, J40_GLOBAL uint32_t *end_bit_out = nullptr,
		J40_LDS uint32_t *ring = nullptr, int32_t ring_stride = 0) {
	LaneOneSection src;
	src.S.start_bit = 8u * sec.byte_off + sec.bit_off; src.S.end_bit = 8u * (sec.byte_off + sec.size);
	src.S.cell_base = cell_base; src.S.block_first = block_first; src.S.nblocks = nblocks; src.S.ev_first = ev_first; src.S.ev_end = ev_end;
	src.taken = false; src.status = 0; src.end_bit = 0;
	decode_hf_sections_lane<SCAN>(f, t, G, src, cols, col_stride, pass, ring, ring_stride);
	if (end_bit_out) *end_bit_out = src.end_bit;
	return src.status;
}

} // namespace j40hip
