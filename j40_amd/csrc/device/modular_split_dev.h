// j40_amd/csrc/device/modular_split_dev.h -- Modular sections whose MA tree looks only at where a sample IS (channel, stream index, row,
// column: properties 0-3; what fast lossless encoders write -- one gradient leaf per channel), decoded in two passes:
//
//   tokens   The stream is parsed front to back by ONE wavefront with every lane in step (wave-uniform, on the scalar unit): the leaf a
//            sample's token is coded with does not depend on any decoded sample, so nothing of the prediction sits on the parse's
//            chain. Out comes the section's flat array of RESIDUAL TOKENS in stream order (the hybrid integers before they are
//            unpacked to signed values) -- which is also exactly the LZ77 window (j40.h:2804-2876: the window holds the decoded integers
//            by their ordinal), so an LZ77 copy of n values is n / 64 vector copies out of that array, not n symbol decodes; values
//            collect in a register across the wavefront and leave 64 at a time.
//   predict  Per channel: v = unpack(token) * multiplier + offset + predictor(neighbours) (j40.h:4222-4231), the one recurrence left.
//            A sample needs W of its own row and NW / N / NE / NEE / NN of the rows above, so 64 consecutive rows go to the 64 lanes of a
//            wavefront, each lane three columns behind the lane above it; the row above reaches a lane through one cross-lane move per
//            step. Bands of 64 rows follow one another (the last two rows of a band wait in LDS for the next).
//
// Same samples, same planes, same error codes as j40__modular_channel (j40.h:4127-4240) decodes them one by one; what the two passes
// have to agree on with it is WHICH error comes first: the parse notes the stream ordinal at which it failed, the prediction the first
// ordinal whose sample leaves the int16 range ("povf"), and the earlier of the two is the section's status (split_status).
//
// Trees that test a decoded neighbour (properties 4+), the weighted predictor and previous-channel properties stay with
// k_modular_sections / k_modular_coop.
#pragma once
#include "modular_dev.h"
#include "hf_uni_dev.h"

namespace j40hip {

// the leaf of a position-only tree for (channel, stream index, y, x)
struct SplitLeaf { int32_t ctx, offset, multiplier, predictor; };
template <bool UNI>
J40_DEV SplitLeaf split_leaf(const DevTreeNode *tree, int32_t cidx, int32_t sidx, int32_t y, int32_t x) {
	const DevTreeNode *n = tree;
	for (;;) {
		const int32_t *w4 = (const int32_t *) n;
		const int32_t prop = uni<UNI>(w4[0]), value = uni<UNI>(w4[1]), a = uni<UNI>(w4[2]), b = uni<UNI>(w4[3]);
		if (prop < 0) { SplitLeaf l = {value, a, b, -1 - prop}; return l; }
		const int32_t val = prop == 0 ? cidx : prop == 1 ? sidx : prop == 2 ? y : x;   // (the host took the tree only with properties 0-3)
		n += val > value ? a : b;
	}
}

// ---- the token pass's symbols on the branch-light bit window of the fast coefficient decoder (UBits, hf_uni_dev.h: absolute position,
// reads that never fail on the padded codestream, the position compared with the section's end once the bits are taken). A lone wavefront
// pays for every instruction it walks past, so what the general reader (entropy_dev.h) spends on refill cases and sticky-error tests
// per read was most of a symbol here. *err receives the first error in the order the general reader would have raised them: "shrt" in
// the token, "iovf", "shrt" in the extra bits; it is left alone once set.

// hybrid integer (j40.h:2313-2334)
template <bool UNI>
J40_DEV int32_t split_hybrid(UBits &b, int32_t token, uint32_t cfg, int32_t max_token, uint32_t end_bit, uint32_t *err) {
	const int32_t split_exp = (int32_t) (cfg & 15), split = 1 << split_exp;
	if (token < split) return token;
	const bool iovf = token > max_token;
	const int32_t tok = iovf ? max_token : token;
	const int32_t msb = (int32_t) ((cfg >> 4) & 15), lsb = (int32_t) ((cfg >> 8) & 15), in_token = msb + lsb;
	const int32_t midbits = split_exp - in_token + ((tok - split) >> in_token);
	if (midbits > b.nbits) ub_refill<UNI>(b);
	const int32_t mid = (int32_t) ub_take(b, midbits);
	if (!*err) *err = iovf ? (uint32_t) ERR_IOVF : ub_position(b) > end_bit ? (uint32_t) ERR_SHRT : 0u;
	const int32_t top = 1 << msb;
	const int32_t lo = tok & ((1 << lsb) - 1), hi = (tok >> lsb) & (top - 1);
	return ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
}
// a prefix-coded token (j40.h:2256-2273); `prefix`: the base DevCluster::table_off indexes
template <bool UNI>
J40_DEV int32_t split_prefix_token(UBits &b, const int32_t *prefix, const ClusterRegs &cl, uint32_t end_bit, uint32_t *err) {
	ub_refill<UNI>(b);
	const int32_t *table = prefix + cl.table_off;
	const uint32_t window = (uint32_t) b.bits & 0xffffu;
	int32_t entry = uni<UNI>(table[window & ((1u << cl.fast_len) - 1)]);
	int32_t used = 0;
	if (entry < 0 && cl.fast_len < cl.max_len) {   // the code is longer than the first table covers: its overflow list
		const int32_t *ovf = table - entry;
		const uint32_t rest = window >> cl.fast_len;
		int32_t code_len, guard = 0;
		do { entry = uni<UNI>(*ovf++); code_len = entry & 15; } while ((uint32_t) ((entry >> 4) & 0xfff) != (rest & ((1u << code_len) - 1)) && ++guard < 32768);
		used = cl.fast_len;
	}
	(void) ub_take(b, used + (entry & 15));
	if (!*err && ub_position(b) > end_bit) *err = ERR_SHRT;
	return entry >> 16;
}
// an rANS token (j40.h:2441-2466); `alias`: the base DevCluster::table_off indexes
template <bool UNI>
J40_DEV int32_t split_ans_token(UBits &b, uint32_t &state, const uint64_t *alias, int32_t log_bucket, const ClusterRegs &cl, uint32_t end_bit, uint32_t *err) {
	ub_refill<UNI>(b);
	if (state == 0) { state = ub_take(b, 16); state |= ub_take(b, 16) << 16; ub_refill<UNI>(b); if (!*err && ub_position(b) > end_bit) *err = ERR_SHRT; }
	const uint32_t idx = state & 0xfff, i = idx >> log_bucket, pos = idx & ((1u << log_bucket) - 1);
	const uint64_t e = uni64<UNI>(alias[cl.table_off + i]);
	const uint32_t elo = (uint32_t) e, ehi = (uint32_t) (e >> 32);
	const bool aliased = pos >= (elo & 0xff);
	const int32_t token = (int32_t) (aliased ? (elo >> 20) & 0xff : i);
	const uint32_t offset = aliased ? (elo >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (ehi >> 9) & 0x1fff;
	state = d * (state >> 12) + offset + pos;
	const bool renorm = state < (1u << 16);
	const uint32_t low = ub_take(b, renorm ? 16 : 0);
	state = renorm ? (state << 16) | low : state;
	if (!*err && ub_position(b) > end_bit) *err = ERR_SHRT;
	return token;
}
// the end of the stream: the rANS state's final value (j40.h:2884-2895), then -- frames that are one section -- the zero padding and
// where the section ends against where the TOC said (j40.h:7796-7803; bits_finish_section)
template <bool UNI>
J40_DEV void split_finish(UBits &b, bool prefix, uint32_t state, uint32_t end_bit, bool check_end, uint32_t declared_end, uint32_t *err) {
	if (*err) return;
	if (!prefix) {
		if (state) { if (state != 0x130000u) { *err = ERR_ANS; return; } }
		else {
			ub_refill<UNI>(b);
			const uint32_t a = ub_take(b, 16); const bool short_a = ub_position(b) > end_bit;
			ub_refill<UNI>(b);
			const uint32_t c = ub_take(b, 16); const bool short_c = ub_position(b) > end_bit;
			*err = short_a ? (uint32_t) ERR_SHRT : a != 0 ? (uint32_t) ERR_ANS : short_c ? (uint32_t) ERR_SHRT : c != 0x13 ? (uint32_t) ERR_ANS : 0u;
			if (*err) return;
		}
	}
	if (check_end) {
		ub_refill<UNI>(b);
		const uint32_t p = ub_position(b), n = (8u - (p & 7u)) & 7u;
		if (ub_take(b, (int32_t) n)) { *err = ERR_PAD0; return; }
		const uint32_t at = (p + n) >> 3;
		if (at < declared_end) *err = ERR_SHRT; else if (at > declared_end) *err = ERR_EXCS;
	}
}

J40_DEV int32_t split_unpack(int32_t token) { return (token & 1) ? -(token / 2 + 1) : token / 2; }   // j40.h:2799

// one sample of the prediction pass; false: it leaves the int16 range (the caller notes the ordinal; the value wraps like a store would)
J40_DEV bool split_sample(int32_t token, const SplitLeaf &l, const ModNeigh &p, int32_t *out) {
	static const ModWP no_wp = ModWP();
	uint32_t err = 0;
	int32_t v = split_unpack(token) * l.multiplier + l.offset;
	v += mod_predict(l.predictor, no_wp, p, &err);
	*out = (int32_t) (int16_t) v;
	return v >= -32768 && v <= 32767;
}

// the section's verdict from the two passes' notes (parse: code and ordinal, 0xffffffff = none; prediction: first overflowing ordinal)
J40_DEV uint32_t split_status(uint32_t parse_code, uint32_t parse_at, uint32_t povf_at) {
	if (povf_at != 0xffffffffu && (!parse_code || povf_at < parse_at)) return ERR_POVF;
	return parse_code;
}

}  // namespace j40hip
