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
