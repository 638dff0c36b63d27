// j40_amd/csrc/modular.hpp -- Modular sub-bitstream on the host: MA tree, header, per-pixel decode,
// inverse transforms. Used for the side streams the reference keeps on the CPU path (LfGlobal,
// LfGroup, raw dequantisation matrices); pass-group Modular streams run in HIP (device/).
//
// Reference behaviour: j40__tree (j40.h:3461), j40__modular_header (3717),
// j40__modular_channel (4127), j40__inverse_rct / j40__inverse_palette (4318, 4402).
#pragma once
#include "entropy.hpp"
#include <memory>

namespace j40hip {

// flattened MA tree node, shared verbatim with the device (16 bytes)
//   branch: prop >= 0 (property index); value = threshold; a / b = offsets (relative to this node)
//           of the child taken when property > threshold / otherwise
//   leaf:   prop = -1 - predictor; value = context id; a = offset; b = multiplier
struct TreeNode { int32_t prop, value, a, b; };

struct WPParams { int8_t p1 = 16, p2 = 10, p3[5] = {7, 7, 7, 0, 0}, w[4] = {13, 12, 12, 12}; };

struct Transform {
	enum Kind { RCT = 0, PALETTE = 1, SQUEEZE = 2 } kind = RCT;
	int32_t begin_c = 0, rct_type = 0;
	int32_t num_c = 0, nb_colours = 0, nb_deltas = 0, d_pred = 0;
	// Squeeze, one step: channels [begin_c, begin_c + num_c) were halved along one axis; their residual channels sit behind
	// them (in_place) or at the end of the channel list (ISO 18181-1 Squeeze; the reference stops at "TODO", j40.h:3812)
	bool horizontal = false, in_place = false;
};


// int16 sample plane (the reference's Main-profile level 5 limits force 16-bit Modular buffers,
// j40.h:1173, 3169)
struct Plane {
	int32_t width = 0, height = 0;
	int8_t hshift = 0, vshift = 0;
	std::vector<int16_t> px;
	bool empty() const { return width <= 0 || height <= 0; }
	void allocate() { px.assign((size_t) (width > 0 ? width : 0) * (size_t) (height > 0 ? height : 0), 0); }
	int16_t *row(int32_t y) { return px.data() + (size_t) y * (size_t) width; }
	const int16_t *row(int32_t y) const { return px.data() + (size_t) y * (size_t) width; }
};

struct Modular {
	bool use_global_tree = false;
	WPParams wp;
	std::vector<Transform> transforms;
	const std::vector<TreeNode> *tree = nullptr;
	const CodeSpec *codespec = nullptr;
	std::vector<TreeNode> own_tree;
	CodeSpec own_codespec;
	std::vector<Plane> channel;
	int32_t nb_meta_channels = 0;
	int32_t dist_mult = 0;
	int32_t bpp = 8;  // image bit depth, needed by the palette transform
};

// reads an MA tree followed by its code spec (j40.h:3461)
void read_tree(BitReader &br, int32_t max_tree_size, int32_t depth_limit, std::vector<TreeNode> *tree, CodeSpec *codespec);
bool tree_uses_wp(const std::vector<TreeNode> &tree);

// channel dimensions must already be set in m->channel
void read_modular_header(BitReader &br, const std::vector<TreeNode> *global_tree, const CodeSpec *global_codespec, Modular *m);
void allocate_modular(Modular *m);
void decode_modular_channel(BitReader &br, Modular &m, CodeState &code, int32_t cidx, int64_t sidx);
void inverse_transforms(Modular &m);
// channel list bookkeeping of one Squeeze step (forward direction, header time): sizes, shifts, residual channels inserted
void apply_squeeze_meta(const Transform &tr, std::vector<Plane> *channel, int32_t *nb_meta);
// the default Squeeze parameter list for a channel list (num_sq = 0 in the bitstream)
void default_squeeze_steps(const std::vector<Plane> &channel, int32_t nb_meta, std::vector<Transform> *out);

// convenience: header, every channel, finish, inverse transforms
void decode_modular_image(BitReader &br, const std::vector<TreeNode> *global_tree, const CodeSpec *global_codespec, int64_t sidx, Modular *m);

} // namespace j40hip
