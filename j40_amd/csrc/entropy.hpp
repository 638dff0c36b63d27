// j40_amd/csrc/entropy.hpp -- entropy code specification: parser (host) and host-side decoder.
//
// The parser builds the tables both the host decoder (LfGlobal / LfGroup / HfGlobal side streams)
// and the HIP kernels (pass-group streams) consume. Reference behaviour: j40__read_code_spec
// (j40.h:2711), j40__cluster_map (2526), j40__ans_table (2601), j40__init_alias_map (2362),
// j40__prefix_code_tree (2049), j40__code (2804), j40__finish_and_free_code (2884).
#pragma once
#include "common.hpp"

namespace j40hip {

struct HybridCfg {
	int8_t split_exp = 0, msb_in_token = 0, lsb_in_token = 0;
	int32_t max_token = 0;
	// packed for the device: bits 0-3 split_exp, 4-7 msb, 8-11 lsb
	uint32_t packed() const { return (uint32_t) split_exp | ((uint32_t) msb_in_token << 4) | ((uint32_t) lsb_in_token << 8); }
};

// one bucket of the rANS alias table, already joined with the two probabilities a decode step
// needs so that a step costs a single table read (device layout, also used by the host decoder)
//   bits  0-7  cutoff            (0..128)
//   bits  8-19 offset            (0..4095)
//   bits 20-27 alias symbol
//   bits 28-40 D[alias symbol]   (0..4096)
//   bits 41-53 D[own symbol]     (0..4096)
using AnsEntry = uint64_t;

struct Cluster {
	HybridCfg cfg;
	// ANS
	std::vector<int16_t> D;
	std::vector<AnsEntry> alias;
	// prefix code: LUT of 1 << fast_len entries followed by overflow entries;
	// entry = symbol << 16 | (code >> fast_len) << 4 | (len - fast_len); negative = -overflow index
	int32_t fast_len = 0, max_len = 0;
	std::vector<int32_t> table;
	std::vector<uint8_t> lengths;   // per symbol, kept for the plan view (oracle builds its own decoder from them)
};

struct CodeSpec {
	int32_t num_dist = 0;
	bool lz77_enabled = false, use_prefix_code = false;
	int32_t min_symbol = 0x7fffffff, min_length = 0x7fffffff;
	int32_t log_alpha_size = 0;
	int32_t num_clusters = 0;
	std::vector<uint8_t> cluster_map;
	HybridCfg lz_len_cfg;
	std::vector<Cluster> clusters;
	bool empty() const { return num_dist == 0; }
};

HybridCfg read_hybrid_cfg(BitReader &br, int32_t log_alpha_size);
void read_cluster_map(BitReader &br, int32_t num_dist, int32_t max_allowed, int32_t *num_clusters, std::vector<uint8_t> *map);
// builds the decoding tables (alias map / prefix LUT) of every cluster from its D / lengths (specs that were not read from a
// bitstream: j40hip_frame_from_vardct_view). Throws DecodeError on distributions the reader would have refused.
void finish_code_spec_tables(CodeSpec *spec);
// num_dist < 0: LZ77 is not allowed for this spec (j40.h:2710)
void read_code_spec(BitReader &br, int32_t num_dist, CodeSpec *spec);

struct CodeState {
	const CodeSpec *spec = nullptr;
	int32_t num_to_copy = 0, copy_pos = 0, num_decoded = 0;
	std::vector<int32_t> window;  // 1 << 20 entries once LZ77 is in use
	uint32_t ans_state = 0;
	explicit CodeState(const CodeSpec *s = nullptr) : spec(s) {}
};

int32_t decode_symbol(BitReader &br, CodeState &code, int32_t ctx, int32_t dist_mult);
void finish_code(BitReader &br, CodeState &code);

// Lehmer-coded permutation (j40.h:5428); returns the raw code (empty = identity)
std::vector<int32_t> read_permutation(BitReader &br, CodeState &code, int32_t size, int32_t skip);
template <typename T> void apply_permutation(T *target, const std::vector<int32_t> &lehmer) {  // j40.h:5460
	for (int32_t x : lehmer) {
		T tmp = target[x];
		memmove(target + 1, target, sizeof(T) * (size_t) x);
		target[0] = tmp;
		++target;
	}
}

} // namespace j40hip
