// j40_amd/csrc/device/lf_decode.hip -- LfGroup sections of VarDCT frames decoded on the device (replaces the
// stream-decoding half of j40__lf_group, j40.h:6722-6790: j40__modular_channel over the LF coefficient image and over the HF
// metadata image). One LfGroup section per wavefront, decoded by all 64 lanes together with the wave-cooperative channel decoder
// (modular_coop_dev.h). The two sub-images are one bit stream without a length in between, so the wavefront goes straight on:
// LF coefficients -> final rANS state -> varblock count -> second Modular header -> HF metadata -> final rANS state.
// What the host keeps: everything in front of the first stream (it knows where that is), and everything after the decode that is
// cheap and irregular -- LF index, varblock placement (frame.cpp, lf_group_finish) -- plus the plan build.
// Only the plain second header is handled here (global tree, default weighted-predictor parameters, no transforms: what VarDCT
// encoders write); any other reports ERR_LFFB and the host decodes that section itself.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "modular_coop_dev.h"
#include "lf_lanes_dev.h"
#include "lf_rows_dev.h"
#include "kernels.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace j40hip {

__global__ void __launch_bounds__(64) k_lf_groups(const DevLfTask *tasks) {
	const uint32_t lane = threadIdx.x;
	const DevLfTask *tp = tasks + blockIdx.x;
	const int32_t w8 = coop_sc(tp->w8), h8 = coop_sc(tp->h8), w64 = coop_sc(tp->w64), h64 = coop_sc(tp->h64);
	const int32_t sidx0 = coop_sc(tp->sidx0), sidx2 = coop_sc(tp->sidx2), nbvb_bits = coop_sc(tp->nbvb_bits);
	const uint32_t info_capacity = (uint32_t) coop_sc((int32_t) tp->info_capacity);
	int16_t *lf_out[3] = {coop_sc_ptr(tp->lf[0]), coop_sc_ptr(tp->lf[1]), coop_sc_ptr(tp->lf[2])};
	int16_t *out_xfromy = coop_sc_ptr(tp->xfromy), *out_bfromy = coop_sc_ptr(tp->bfromy), *out_info = coop_sc_ptr(tp->info), *out_sharp = coop_sc_ptr(tp->sharp);
	DevLfResult *result = coop_sc_ptr(tp->result);
	const uint8_t *codestream = coop_sc_ptr(tp->codestream);
	const CoopTreeRegs t = coop_load_tree(coop_sc_ptr(tp->tree), lane);
	const int32_t log_bucket = 12 - coop_sc(tp->log_alpha_size);
	const CoopConstU64 alias = (CoopConstU64) coop_sc_ptr(tp->alias);
	CoopBits b;
	coop_bits_init(b, codestream, (uint32_t) coop_sc((int32_t) tp->byte_off), (uint32_t) coop_sc((int32_t) tp->size), (uint32_t) coop_sc((int32_t) tp->bit_off), lane);
	uint32_t state = 0, err = 0, status = 0;
	int32_t nb_varblocks = 0;
	// the LF coefficient image
	for (int32_t c = 0; c < 3 && !b.err && !err; ++c)
		coop_decode_channel<0, true>(b, state, err, t, alias, log_bucket, c, sidx0, c == 0 ? lf_out[0] : c == 1 ? lf_out[1] : lf_out[2], w8, w8, h8, nullptr, 0, lane);
	status = b.err ? b.err : err;
	if (!status) status = coop_finish_code(b, state, lane);
	if (!status) {
		nb_varblocks = (int32_t) coop_take(b, nbvb_bits, lane) + 1;   // j40.h:6601
		const uint32_t header = coop_take(b, 4, lane);                // use_global_tree = 1, default wp = 1, no transforms (j40.h:3717-3760)
		if (b.err) status = b.err;
		else if (header != 3u) status = ERR_LFFB;
		else if (2 * (uint32_t) nb_varblocks > info_capacity) status = ERR_LFFB;   // (more varblocks than cells: the host reports it)
	}
	if (!status) {   // the HF metadata image
		state = 0;
		coop_decode_channel<0, true>(b, state, err, t, alias, log_bucket, 0, sidx2, out_xfromy, w64, w64, h64, nullptr, 0, lane);
		if (!b.err && !err) coop_decode_channel<0, true>(b, state, err, t, alias, log_bucket, 1, sidx2, out_bfromy, w64, w64, h64, nullptr, 0, lane);
		if (!b.err && !err) coop_decode_channel<0, true>(b, state, err, t, alias, log_bucket, 2, sidx2, out_info, nb_varblocks, nb_varblocks, 2, nullptr, 0, lane);
		if (!b.err && !err) coop_decode_channel<0, true>(b, state, err, t, alias, log_bucket, 3, sidx2, out_sharp, w8, w8, h8, nullptr, 0, lane);
		status = b.err ? b.err : err;
		if (!status) status = coop_finish_code(b, state, lane);
	}
	if (lane == 0) { result->status = status; result->nb_varblocks = nb_varblocks; }
}

// One LfGroup section per LANE (lf_lanes_dev.h); a wavefront takes the sections of several frames (DevLfWave), each frame's tree and
// small code tables staged in LDS, every lane pointing at its own frame's copy. ALIAS_LDS: the alias tables as well.
template <bool ALIAS_LDS>
__global__ void __launch_bounds__(64) k_lf_lanes(const DevLfLaneSet *sets, const DevLfWave *waves) {
	extern __shared__ __attribute__((aligned(16))) uint8_t lfl_lds[];
	const J40_GLOBAL DevLfWave &wv = ((const J40_GLOBAL DevLfWave *) waves)[blockIdx.x];
	const int32_t lane = threadIdx.x, num_parts = wv.num_parts;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	J40_LDS uint8_t *lds = (J40_LDS uint8_t *) lfl_lds;
	// this lane's section and tables
	const J40_GLOBAL DevLfTask *task = nullptr;
	LfLaneTablesT<ALIAS_LDS> T;
	T.ctx_map = nullptr; T.cluster_cfg = nullptr; T.alias = nullptr; T.log_alpha = 5; T.log_bucket = 7;
	LfLaneFrame F;
	F.tree = nullptr; F.uses = 0;
	uint32_t at = 0; int32_t lane0 = 0;
	for (int32_t p = 0; p < num_parts; ++p) {
		const J40_GLOBAL DevLfLaneSet &set = ((const J40_GLOBAL DevLfLaneSet *) sets)[wv.part[p].set];
		const int32_t first = wv.part[p].first_task, count = wv.part[p].count;
		const int32_t num_nodes = set.num_nodes, num_dist = set.num_dist, num_clusters = set.num_clusters, log_alpha = set.log_alpha;
		J40_LDS int32_t *l_tree = (J40_LDS int32_t *) (lds + at);
		J40_LDS uint8_t *l_map = lds + at + align16(16u * (uint32_t) num_nodes);
		J40_LDS uint32_t *l_cfg = (J40_LDS uint32_t *) (l_map + align16((uint32_t) num_dist));
		J40_LDS uint64_t *l_alias = (J40_LDS uint64_t *) ((J40_LDS uint8_t *) l_cfg + align16(4u * (uint32_t) num_clusters));
		{
			const J40_GLOBAL int32_t *tsrc = (const J40_GLOBAL int32_t *) set.tree;
			for (int32_t i = lane; i < 4 * num_nodes; i += 64) l_tree[i] = tsrc[i];
			const J40_GLOBAL uint8_t *msrc = (const J40_GLOBAL uint8_t *) set.ctx_map;
			for (int32_t i = lane; i < num_dist; i += 64) l_map[i] = msrc[i];
			const J40_GLOBAL uint32_t *csrc = (const J40_GLOBAL uint32_t *) set.cluster_cfg;
			for (int32_t i = lane; i < num_clusters; i += 64) l_cfg[i] = csrc[i];
			if constexpr (ALIAS_LDS) {
				const J40_GLOBAL uint64_t *asrc = (const J40_GLOBAL uint64_t *) set.alias;
				for (int32_t i = lane; i < (num_clusters << log_alpha); i += 64) l_alias[i] = asrc[i];
			}
		}
		if (lane >= lane0 && lane < lane0 + count) {
			task = (const J40_GLOBAL DevLfTask *) set.tasks + (first + lane - lane0);
			T.ctx_map = l_map; T.cluster_cfg = l_cfg; T.log_alpha = log_alpha; T.log_bucket = 12 - log_alpha;
			if constexpr (ALIAS_LDS) T.alias = l_alias; else T.alias = (const J40_GLOBAL uint64_t *) set.alias;
			F.tree = (const J40_LDS DevTreeNode *) l_tree; F.uses = set.uses;
		}
		lane0 += count;
		at += align16(set.lds_bytes) + (ALIAS_LDS ? 8u * ((uint32_t) num_clusters << log_alpha) : 0u);
	}
	__syncthreads();
	const bool active = task != nullptr;
	if (!active) {   // (something valid to point at)
		const J40_GLOBAL DevLfLaneSet &set = ((const J40_GLOBAL DevLfLaneSet *) sets)[wv.part[0].set];
		task = (const J40_GLOBAL DevLfTask *) set.tasks + wv.part[0].first_task;
	}
	const J40_GLOBAL DevLfTask &t = *task;
	LfLane L;
	lf_lane_init(L, t);
	if (!active) { L.chan = 7; L.setup = false; }
	while (__builtin_amdgcn_ballot_w64(!lf_lane_done(L))) lf_lane_step(L, t, F, T);
	if (active) { J40_GLOBAL DevLfResult *r = (J40_GLOBAL DevLfResult *) t.result; r->status = L.err; r->nb_varblocks = L.nb_varblocks; }
}

// the pieces of rows the lanes' last steps completed, copied out by the whole wavefront: lane i takes samples i, i + 64, ... of
// each piece, so a piece leaves as runs of 128 consecutive bytes
J40_DEV void lf_row_flush_wave(LfRowLane &L, int32_t lane) {
	uint64_t need = __builtin_amdgcn_ballot_w64(L.flush_n > 0);
	const uint64_t dst_bits = (uint64_t) (uintptr_t) L.flush_dst;
	const uint32_t win_bits = (uint32_t) (uintptr_t) L.win;
	while (need) {
		const int32_t j = (int32_t) __builtin_ctzll(need);
		need &= need - 1;
		const int32_t n = __builtin_amdgcn_readlane(L.flush_n, j);
		const uint64_t d = (uint64_t) (uint32_t) __builtin_amdgcn_readlane((int32_t) (uint32_t) dst_bits, j) | ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int32_t) (uint32_t) (dst_bits >> 32), j) << 32);
		J40_GLOBAL int16_t *dst = (J40_GLOBAL int16_t *) (uintptr_t) d;
		const J40_LDS int16_t *src = (const J40_LDS int16_t *) (uintptr_t) (uint32_t) __builtin_amdgcn_readlane((int32_t) win_bits, j);
		for (int32_t i = lane; i < n; i += 64) dst[i] = src[i];
	}
	L.flush_n = 0;
}

// One LfGroup section per LANE, tree + fast entries (lf_rows_dev.h) + row windows in LDS. A wavefront takes the sections of the frames
// pack_lf_row_waves gave it; LDS: per part the staged tree and the fast entries made from the frame's alias tables, then one window
// per section. A part's sections go to its lanes in list order (the host lists them by decreasing size).
__global__ void __launch_bounds__(64) k_lf_rows(const DevLfLaneSet *sets, const DevLfWave *waves, int32_t raw) {
	extern __shared__ __attribute__((aligned(16))) uint8_t lfr_lds[];
	const J40_GLOBAL DevLfWave &wv = ((const J40_GLOBAL DevLfWave *) waves)[blockIdx.x];
	const int32_t lane = threadIdx.x, num_parts = wv.num_parts;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	J40_LDS uint8_t *lds = (J40_LDS uint8_t *) lfr_lds;
	// the wavefront's sections, part after part, are numbered 0, 1, ...: lane l takes section l
	const J40_GLOBAL DevLfTask *task = nullptr;
	LfRowTables T;
	T.tree = nullptr; T.fast = nullptr; T.alias = nullptr; T.log_alpha = 5; T.log_bucket = 7; T.uses = 0;
	uint32_t at = 0; int32_t section0 = 0;
	for (int32_t p = 0; p < num_parts; ++p) {
		const J40_GLOBAL DevLfLaneSet &set = ((const J40_GLOBAL DevLfLaneSet *) sets)[wv.part[p].set];
		const int32_t first = wv.part[p].first_task, count = wv.part[p].count;
		const int32_t num_nodes = set.num_nodes, num_clusters = set.num_clusters, log_alpha = set.log_alpha;
		J40_LDS int32_t *l_tree = (J40_LDS int32_t *) (lds + at);
		J40_LDS LfFastQuad *l_fast = (J40_LDS LfFastQuad *) (lds + at + align16(16u * (uint32_t) num_nodes));
		{
			const J40_GLOBAL int32_t *tsrc = (const J40_GLOBAL int32_t *) set.tree;
			const J40_GLOBAL uint8_t *msrc = (const J40_GLOBAL uint8_t *) set.ctx_map;
			const J40_GLOBAL uint32_t *csrc = (const J40_GLOBAL uint32_t *) set.cluster_cfg;
			for (int32_t i = lane; i < num_nodes; i += 64) {
				const int32_t prop = tsrc[4 * i]; int32_t value = tsrc[4 * i + 1];
				if (prop < 0) { const uint32_t cl = msrc[value]; value = lf_rows_leaf_word(cl, csrc[cl]); }   // (the host checked value < num_dist)
				l_tree[4 * i] = prop; l_tree[4 * i + 1] = value; l_tree[4 * i + 2] = tsrc[4 * i + 2]; l_tree[4 * i + 3] = tsrc[4 * i + 3];
			}
			const J40_GLOBAL uint64_t *asrc = (const J40_GLOBAL uint64_t *) set.alias;
			for (int32_t i = lane; i < (num_clusters << log_alpha); i += 64) l_fast[i] = lf_rows_fast_entry(asrc[i], (uint32_t) i & ((1u << log_alpha) - 1u), csrc[i >> log_alpha]);
		}
		if (lane >= section0 && lane < section0 + count) {
			task = (const J40_GLOBAL DevLfTask *) set.tasks + (first + lane - section0);
			T.tree = (const J40_LDS DevTreeNode *) l_tree; T.fast = (const J40_LDS uint32_t *) l_fast; T.alias = (const J40_GLOBAL uint64_t *) set.alias;
			T.log_alpha = log_alpha; T.log_bucket = 12 - log_alpha; T.uses = set.uses | (raw ? (uint32_t) LF_USES_RAW : 0u);
		}
		section0 += count;
		at += lf_rows_table_bytes(num_nodes, num_clusters, log_alpha);
	}
	J40_LDS int16_t *wins = (J40_LDS int16_t *) (lds + at);
	__syncthreads();
	const bool active = task != nullptr;
	if (!active) {   // (something valid to point at)
		const J40_GLOBAL DevLfLaneSet &set = ((const J40_GLOBAL DevLfLaneSet *) sets)[wv.part[0].set];
		task = (const J40_GLOBAL DevLfTask *) set.tasks + wv.part[0].first_task;
		T.tree = (const J40_LDS DevTreeNode *) lds; T.fast = (const J40_LDS uint32_t *) lds; T.alias = (const J40_GLOBAL uint64_t *) set.alias;
	}
	const J40_GLOBAL DevLfTask &t = *task;
	LfRowLane L;
	lf_row_init(L, t, wins + (active ? lane : 0) * LF_ROW_PITCH);   // (a lane without a section never writes its window)
	if (!active) { L.chan = 7; L.setup = false; }   // (the first general step finds it finished)
	// Every pass: the lanes run through their stretches of plain samples (lf_row_run_plain, in the instantiation that covers what
	// their channels need), until some live lane is at a channel start, a row's end or in a channel of another form; those lanes
	// then take the general step (which says how long the lane's next run is), finished rows leave, and the needs are taken again
	uint32_t need = LF_NEED_ALL;
	for (;;) {
		lf_row_run_plain(L, T, need);
		const bool plain = L.plain_left > 0;
		if (!plain) lf_row_step(L, t, T);
		if (__builtin_amdgcn_ballot_w64(L.flush_n > 0)) lf_row_flush_wave(L, lane);
		if (!__builtin_amdgcn_ballot_w64(L.live)) break;
		const uint32_t mine = lf_plain_needs(L);
		need = 0;
		for (uint32_t bit = 1; bit <= 16; bit <<= 1) need |= __builtin_amdgcn_ballot_w64((mine & bit) != 0) ? bit : 0u;
	}
	if (active) {
		J40_GLOBAL DevLfResult *r = (J40_GLOBAL DevLfResult *) t.result; r->status = L.err; r->nb_varblocks = L.nb_varblocks;
		if (raw) { r->raw_mask = L.raw_mask; r->stopped_at = L.stopped_at; }   // (for k_lf_predict, which follows on the stream)
	}
}

// The predictions of the channels k_lf_rows left as residuals (lf_rows_dev.h, RAW channels): one wavefront per section, channel after
// channel in stream order. 64 rows at a time wait in LDS (copied in and out in runs of consecutive addresses: with every lane on a
// row of its own in global memory a load or store is 64 cache lines, and the kernel took 18 ms per launch of 256 frames whether one
// wavefront or three shared a SIMD) and go to the 64 lanes, each lane three columns behind the lane above it, which hands it the row
// above through one cross-lane move per step (k_modular_predict's scheme, modular_split.hip); the band's last row stays behind for
// the next band's first lane. The step is one basic block: positions outside the row read a clamped slot and write the row's spare
// slot. A sample is read (as a residual) by the lane that replaces it, once. Workgroup b: section b / W of wavefront b % W of the
// launch's list of W.
template <int PRED>
__device__ __forceinline__ bool lf_predict_band(J40_LDS int16_t *tile, const J40_LDS int32_t *above, int32_t lane, int32_t y, int32_t cw, int32_t width) {
	J40_LDS int16_t *row = tile + lane * LF_ROW_PITCH;
	int32_t r_nww = 0, r_nw = 0, r_n = 0, r_ne = 0, r_nee = 0, c_w = 0, c_ww = 0;
	bool povf = false;
	const int32_t last = width > 0 ? width - 1 : 0;
	int32_t next = row[mod_min(mod_max(-3 * lane - 2, 0), last)];   // the residual of the step to come (read a step ahead of its use)
	const int32_t steps = cw + 3 * 63 + 3;
	for (int32_t tstep = 0; tstep < steps; ++tstep) {
		const int32_t x = tstep - 3 * lane - 2;   // (every lane starts two columns early: its registers fill with the row above at 0, 1, 2)
		const int32_t res = next;
		next = row[mod_min(mod_max(x + 1, 0), last)];
		int32_t in_nee = __builtin_amdgcn_ds_bpermute((lane > 0 ? lane - 1 : 0) << 2, c_w);   // the row above at x + 2: what the lane above computed a step ago
		const int32_t from_above = above[mod_min(mod_max(x + 2, 0), LF_ROW_WIN)];
		in_nee = lane == 0 ? (x + 2 >= 0 && x + 2 < cw ? from_above : 0) : in_nee;
		r_nww = r_nw; r_nw = r_n; r_n = r_ne; r_ne = r_nee; r_nee = in_nee;   // now centred on x
		const bool valid = x >= 0 && x < width;
		const int32_t v = lf_predict_value(res, PRED, x, y, cw, c_w, c_ww, r_nww, r_nw, r_n, r_ne);
		povf |= valid && (v < -32768 || v > 32767);
		row[valid ? x : LF_ROW_WIN] = (int16_t) v;   // (slot 256 of a lane's 258 is nobody's sample)
		c_ww = valid ? c_w : c_ww; c_w = valid ? v : c_w;
	}
	return povf;
}
__global__ void __launch_bounds__(64) k_lf_predict(const DevLfLaneSet *sets, const DevLfWave *waves) {
	__shared__ int16_t tile_lds[65 * LF_ROW_PITCH];   // (a row more than the band: the copy in goes two rows at a time)
	__shared__ int32_t above_lds[LF_ROW_WIN + 4];
	J40_LDS int16_t *tile = (J40_LDS int16_t *) tile_lds;
	J40_LDS int32_t *above = (J40_LDS int32_t *) above_lds;
	// (section-major: the workgroups that have a section come first whatever the wavefronts' fill -- with twelve sections per wavefront
	// of k_lf_rows and the wavefront's 64 workgroups side by side, half of the XCDs got two sections for the others' one: 9.0 ms against 4.6)
	const uint32_t num_waves = gridDim.x >> 6;
	const J40_GLOBAL DevLfWave &wv = ((const J40_GLOBAL DevLfWave *) waves)[blockIdx.x % num_waves];
	const int32_t my_section = (int32_t) (blockIdx.x / num_waves), lane = threadIdx.x;
	const J40_GLOBAL DevLfTask *task = nullptr;
	int32_t section0 = 0;
	for (int32_t p = 0; p < wv.num_parts; ++p) {
		const int32_t count = wv.part[p].count;
		if (my_section >= section0 && my_section < section0 + count) task = (const J40_GLOBAL DevLfTask *) ((const J40_GLOBAL DevLfLaneSet *) sets)[wv.part[p].set].tasks + (wv.part[p].first_task + my_section - section0);
		section0 += count;
	}
	if (!task) return;
	const J40_GLOBAL DevLfTask &t = *task;
	J40_GLOBAL DevLfResult *r = (J40_GLOBAL DevLfResult *) t.result;
	const uint32_t raw_mask = r->raw_mask, stopped_at = r->stopped_at;
	const int32_t nb_varblocks = r->nb_varblocks;
	bool povf = false;
	for (int32_t chan = 0; chan < 7 && !povf; ++chan) {
		const int32_t nib = (int32_t) ((raw_mask >> (4 * chan)) & 15u);
		if (nib < 2) continue;   // samples already, or nothing to add (predictor 0)
		int32_t cw, chh;
		J40_GLOBAL int16_t *plane = lf_channel_plane(t, chan, nb_varblocks, &cw, &chh);
		const int32_t limit = lf_channel_complete(stopped_at, chan, cw, chh);
		if (limit <= 0 || cw > LF_ROW_WIN) continue;   // (rows wider than the window are never left as residuals with something to predict)
		const int32_t rows = (limit + cw - 1) / cw;
		for (int32_t y0 = 0; y0 < rows; y0 += 64) {
			__syncthreads();
			for (int32_t x = lane; x <= LF_ROW_WIN; x += 64) above[x] = y0 > 0 && x < cw ? (int32_t) tile[63 * LF_ROW_PITCH + x] : 0;   // the band before left its last row
			__syncthreads();
			J40_GLOBAL int16_t *band = plane + (size_t) y0 * (size_t) cw;
			const int32_t n = mod_min(limit - y0 * cw, 64 * cw);
			// (row by row, a lane every 64th sample of a row, eight loads asked for before the first is stored: a loop of one load and one
			// store per turn waits for global memory 256 times a band, longer than the band's prediction takes)
			for (int32_t ry = 0; ry * cw < n; ry += 2) {
				int16_t a[4], b[4];
#pragma unroll
				for (int32_t k = 0; k < 4; ++k) {
					const int32_t x = lane + 64 * k, i = ry * cw + x;
					a[k] = x < cw && i < n ? band[i] : (int16_t) 0;
					b[k] = x < cw && i + cw < n ? band[i + cw] : (int16_t) 0;
				}
#pragma unroll
				for (int32_t k = 0; k < 4; ++k) {
					const int32_t x = lane + 64 * k;
					if (x < cw) { tile[ry * LF_ROW_PITCH + x] = a[k]; tile[(ry + 1) * LF_ROW_PITCH + x] = b[k]; }   // (ry + 1 <= 64: the tile has 64 rows and the window's two spare slots more)
				}
			}
			__syncthreads();
			const int32_t y = y0 + lane;
			const int32_t width = y < rows ? mod_min(cw, limit - y * cw) : 0;   // (the section's last row may be a piece of one)
			bool bad;
			switch (nib - 1) {
			case 1: bad = lf_predict_band<1>(tile, above, lane, y, cw, width); break;
			case 2: bad = lf_predict_band<2>(tile, above, lane, y, cw, width); break;
			case 3: bad = lf_predict_band<3>(tile, above, lane, y, cw, width); break;
			case 4: bad = lf_predict_band<4>(tile, above, lane, y, cw, width); break;
			case 5: bad = lf_predict_band<5>(tile, above, lane, y, cw, width); break;
			case 7: bad = lf_predict_band<7>(tile, above, lane, y, cw, width); break;
			case 8: bad = lf_predict_band<8>(tile, above, lane, y, cw, width); break;
			case 9: bad = lf_predict_band<9>(tile, above, lane, y, cw, width); break;
			case 10: bad = lf_predict_band<10>(tile, above, lane, y, cw, width); break;
			case 11: bad = lf_predict_band<11>(tile, above, lane, y, cw, width); break;
			default: bad = lf_predict_band<12>(tile, above, lane, y, cw, width); break;
			}
			povf |= bad;
			__syncthreads();
			for (int32_t ry = 0; ry * cw < n; ++ry) {
#pragma unroll
				for (int32_t k = 0; k < 4; ++k) {
					const int32_t x = lane + 64 * k, i = ry * cw + x;
					if (x < cw && i < n) band[i] = tile[ry * LF_ROW_PITCH + x];
				}
			}
		}
		povf = __builtin_amdgcn_ballot_w64(povf) != 0;   // (channels follow one another in the stream: the first one with such a sample decides)
	}
	if (lane == 0) {
		if (povf && r->status != (uint32_t) ERR_LFFB) r->status = ERR_POVF;   // (before the place the lane stopped: only such samples were looked at; a lane that gave up -- "lffb" -- may have left garbage there, and the host decodes its section anyway)
		r->raw_mask = 0; r->stopped_at = 0;   // (the words are the plan build's from here on: DevLfSlot)
	}
}

// J40HIP_LF_ALIAS_LDS=1: the alias tables staged in LDS too (frames whose tables do not fit: the host decodes their sections)
bool lf_lanes_alias_in_lds() {
	static const bool v = [] { const char *e = getenv("J40HIP_LF_ALIAS_LDS"); return e && atoi(e) != 0; }();
	return v;
}

// packs the sections of `sets` into wavefronts (host side): fills `waves`, returns the LDS bytes a wavefront needs at most
uint32_t pack_lf_waves(const DevLfLaneSet *sets_host, int32_t num_sets, std::vector<DevLfWave> *waves, const std::vector<int32_t> *only) {
	// (J40HIP_LF_LDS_KB: the LDS a wavefront's frames may take together -- with the alias tables in LDS, 30 keeps it to one 8K frame per
	// wavefront and two such workgroups beside a coefficient decoder's 99 KB on a compute unit)
	static const uint32_t budget = [] { const char *e = getenv("J40HIP_LF_LDS_KB"); return (e && atoi(e) > 0 ? (uint32_t) atoi(e) : 56u) * 1024u; }();
	uint32_t most = 0, used = 0; int32_t lanes = 0;
	DevLfWave cur; memset(&cur, 0, sizeof cur);
	auto flush = [&] { if (cur.num_parts) { waves->push_back(cur); most = std::max(most, used); } memset(&cur, 0, sizeof cur); used = 0; lanes = 0; };
	for (int32_t k = 0; k < (only ? (int32_t) only->size() : num_sets); ++k) {
		const int32_t i = only ? (*only)[(size_t) k] : k;   // (`only`: just these sets -- the ones k_lf_rows could not take)
		const uint32_t need = ((sets_host[i].lds_bytes + 15u) & ~15u) + (lf_lanes_alias_in_lds() ? 8u * ((uint32_t) sets_host[i].num_clusters << sets_host[i].log_alpha) : 0u);
		for (int32_t first = 0; first < sets_host[i].ntasks; ) {
			if (lanes >= 64 || cur.num_parts >= LF_WAVE_PARTS || (cur.num_parts && used + need > budget)) flush();
			const int32_t count = std::min(64 - lanes, sets_host[i].ntasks - first);
			cur.part[cur.num_parts].set = i; cur.part[cur.num_parts].first_task = first; cur.part[cur.num_parts].count = count; ++cur.num_parts;
			used += need; lanes += count; first += count;
		}
	}
	flush();
	return most;
}

void launch_lf_lanes(const DevLfLaneSet *sets, const DevLfWave *waves, int32_t num_waves, uint32_t lds_bytes, hipStream_t stream, hipEvent_t started, hipEvent_t stopped) {
	if (num_waves <= 0) return;
	static bool configured = false;
	if (!configured) { (void) hipFuncSetAttribute((const void *) k_lf_lanes<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); (void) hipFuncSetAttribute((const void *) k_lf_lanes<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); configured = true; }
	if (lf_lanes_alias_in_lds()) hipLaunchKernelGGL(k_lf_lanes<true>, dim3((unsigned) num_waves), dim3(64), lds_bytes, stream, sets, waves);
	else if (started || stopped) hipExtLaunchKernelGGL(k_lf_lanes<false>, dim3((unsigned) num_waves), dim3(64), lds_bytes, stream, started, stopped, 0, sets, waves);
	else hipLaunchKernelGGL(k_lf_lanes<false>, dim3((unsigned) num_waves), dim3(64), lds_bytes, stream, sets, waves);
}

// J40HIP_LF_KERNEL=lanes: the older decoder (alias entries and rows in global memory) for every launch; default: k_lf_rows, for
// every frame whose tables fit its LDS budget
static int lf_rows_mode() {
	static const int v = [] { const char *e = getenv("J40HIP_LF_KERNEL"); return e && strcmp(e, "lanes") == 0 ? 0 : 1; }();
	return v;
}
bool lf_rows_enabled() { return lf_rows_mode() != 0; }

// packs the sections of `sets` into wavefronts of k_lf_rows (64 lanes, a section each); returns the LDS bytes a wavefront
// needs at most; the sets whose tables do not fit are left out and listed in `oversized` (k_lf_lanes takes them; without the list: 0 and no wavefronts). J40HIP_LF_ROWS_LDS_KB: what a
// wavefront's tables and windows may take together (default 48: one 8K frame of seven clusters x 256 buckets -- 28.5 KB of tables + 12 windows =
// 35 KB -- so that such a workgroup still fits beside the coefficient decoder's 99 KB on a compute unit)
uint32_t pack_lf_row_waves(const DevLfLaneSet *sets_host, int32_t num_sets, std::vector<DevLfWave> *waves, std::vector<int32_t> *oversized) {
	static const uint32_t budget = [] { const char *e = getenv("J40HIP_LF_ROWS_LDS_KB"); return (e && atoi(e) > 0 ? (uint32_t) atoi(e) : 48u) * 1024u; }();
	const uint32_t win_bytes = 2u * LF_ROW_PITCH;
	uint32_t most = 0, used = 0; int32_t lanes = 0;
	DevLfWave cur; memset(&cur, 0, sizeof cur);
	auto flush = [&] { if (cur.num_parts) { waves->push_back(cur); most = std::max(most, used); } memset(&cur, 0, sizeof cur); used = 0; lanes = 0; };
	for (int32_t i = 0; i < num_sets; ++i) {
		const uint32_t tables = lf_rows_table_bytes(sets_host[i].num_nodes, sets_host[i].num_clusters, sets_host[i].log_alpha);
		// (a frame whose tree and alias tables exceed a wavefront's LDS goes to k_lf_lanes -- that frame alone, not the launch's other frames)
		if (tables + win_bytes > 60u * 1024u) { if (oversized) { oversized->push_back(i); continue; } waves->clear(); return 0; }
		for (int32_t first = 0; first < sets_host[i].ntasks; ) {
			if (lanes >= 64 || cur.num_parts >= LF_WAVE_PARTS || (cur.num_parts && used + tables + win_bytes > budget)) flush();
			// as many of the frame's sections as fit the budget (a wavefront's first frame may exceed it, up to the 60 KB a workgroup asks for at most)
			const uint32_t limit = cur.num_parts ? budget : std::min(60u * 1024u, std::max(budget, tables + win_bytes));
			const int32_t room = std::max(1, (int32_t) ((limit - used - tables) / win_bytes));
			const int32_t count = std::min(std::min(64 - lanes, sets_host[i].ntasks - first), room);
			cur.part[cur.num_parts].set = i; cur.part[cur.num_parts].first_task = first; cur.part[cur.num_parts].count = count; ++cur.num_parts;
			used += tables + (uint32_t) count * win_bytes; lanes += count; first += count;
		}
	}
	flush();
	return most;
}

// J40HIP_LF_RAW=0: every channel predicted by the lane that parses it, as before round 6 (A/B runs, tests)
static bool lf_rows_raw() {
	static const bool v = [] { const char *e = getenv("J40HIP_LF_RAW"); return !(e && atoi(e) == 0 && e[0] != 0); }();
	return v;
}
// k_lf_rows, then -- when it leaves leaf-only channels as residuals -- k_lf_predict; `started` / `stopped`: the device's clock before
// the first and after the last of them
void launch_lf_rows(const DevLfLaneSet *sets, const DevLfWave *waves, int32_t num_waves, uint32_t lds_bytes, hipStream_t stream, hipEvent_t started, hipEvent_t stopped) {
	if (num_waves <= 0) return;
	static bool configured = false;
	if (!configured) { (void) hipFuncSetAttribute((const void *) k_lf_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); configured = true; }
	const int32_t raw = lf_rows_raw() ? 1 : 0;
	hipEvent_t rows_stopped = raw ? nullptr : stopped;
	if (started || rows_stopped) hipExtLaunchKernelGGL(k_lf_rows, dim3((unsigned) num_waves), dim3(64), lds_bytes, stream, started, rows_stopped, 0, sets, waves, raw);
	else hipLaunchKernelGGL(k_lf_rows, dim3((unsigned) num_waves), dim3(64), lds_bytes, stream, sets, waves, raw);
	if (raw) {
		if (stopped) hipExtLaunchKernelGGL(k_lf_predict, dim3((unsigned) num_waves * 64u), dim3(64), 0, stream, nullptr, stopped, 0, sets, waves);
		else hipLaunchKernelGGL(k_lf_predict, dim3((unsigned) num_waves * 64u), dim3(64), 0, stream, sets, waves);
	}
}

void launch_lf_groups(const DevLfTask *tasks, int32_t num_tasks, hipStream_t stream) {
	if (num_tasks > 0) hipLaunchKernelGGL(k_lf_groups, dim3((unsigned) num_tasks), dim3(64), 0, stream, tasks);
}

} // namespace j40hip
