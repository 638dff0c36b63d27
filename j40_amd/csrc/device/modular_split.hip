// j40_amd/csrc/device/modular_split.hip -- Modular sections with a position-only MA tree in two passes (modular_split_dev.h says why and how):
//
//   k_modular_tokens    one wavefront per section: the section's stream parsed to its flat array of residual tokens (= the LZ77 window);
//                       prefix codes or rANS, LZ77 runs as vector copies (replaces the entropy half of j40__modular_channel,
//                       j40.h:4127-4240, as j40__code drives it, j40.h:2804-2876)
//   k_modular_predict   one wavefront per (section, channel): prediction + store, 64 rows at a time, each lane three columns behind the
//                       lane above (the other half of j40.h:4222-4231)
//   k_modular_split_status   one lane per section: the two passes' notes -> the section's 4-char code
//
// What this is for: BASELINE config 1 -- one 256 x 256 RGBA section, prefix codes + LZ77, a gradient leaf -- took 272 ms in
// k_modular_sections (a wavefront walking tree, neighbours, predictor and entropy decoder per sample, 1 us each); the reference's single
// core needs 4 ms. Roofline: neither pass is near HBM (4 B written + 4 B read per sample for the tokens, 2 B written for the sample);
// the parse is one wavefront's dependent chain per symbol.
#include <hip/hip_runtime.h>
#include "modular_split_dev.h"
#include "kernels.h"

namespace j40hip {


enum { SPLIT_CHUNK_WORDS = 1024 };   // the token pass's window on the codestream in LDS (k_modular_tokens, the 64-at-a-time mode)
// J40HIP_SPLIT_NO_FAST=1: every symbol through the scalar decoder (the sixty-four-at-a-time mode off, for comparison)
__device__ bool g_split_no_fast = false;
__device__ __forceinline__ bool split_no_fast() { return g_split_no_fast; }

template <bool IN_LDS>
__global__ void __launch_bounds__(64) k_modular_tokens(DevModPlan plan, int32_t first_section) {
	extern __shared__ __attribute__((aligned(16))) uint8_t split_lds[];
	const int32_t lane = threadIdx.x;
	const int32_t s = first_section + (int32_t) blockIdx.x;
	const DevModSection sec = plan.sections[s];
	if (!sec.split) return;
	uint32_t *note = plan.split_state + 3 * (size_t) s;   // parse code, parse ordinal, first overflowing ordinal
	if (sec.preset_status) { if (lane == 0) { note[0] = sec.preset_status; note[1] = 0; note[2] = 0xffffffffu; } return; }
	const DevCodeSpec &spec = plan.spec[sec.spec_idx];
	ModTables t = mod_tables_in_hbm(plan, s);
	if (IN_LDS) {   // the tree, the context map, the clusters and the code's tables in LDS (as k_modular_sections stages them)
		auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
		const int32_t tree_nodes = sec.tree_nodes;
		uint32_t off = 0;
		DevTreeNode *l_tree = (DevTreeNode *) split_lds; off = align16((uint32_t) tree_nodes * (uint32_t) sizeof(DevTreeNode));
		uint32_t *l_map = (uint32_t *) (split_lds + off); off = align16(off + (uint32_t) spec.num_dist + 4);
		DevCluster *l_clusters = (DevCluster *) (split_lds + off); off = align16(off + (uint32_t) spec.num_clusters * (uint32_t) sizeof(DevCluster));
		uint8_t *l_tab = split_lds + off;
		{ const uint4 *src = (const uint4 *) (plan.tree + sec.tree_off); uint4 *dst = (uint4 *) l_tree; for (int32_t i = lane; i < tree_nodes; i += 64) dst[i] = src[i]; }
		{ const uint32_t *src = (const uint32_t *) (plan.pool_u8 + spec.cluster_map_off); for (int32_t i = lane; i < (spec.num_dist + 3) / 4; i += 64) l_map[i] = src[i]; }
		const DevCluster *csrc = plan.clusters + spec.cluster_off;
		const uint32_t base_off = csrc[0].table_off;
		for (int32_t i = lane; i < spec.num_clusters; i += 64) { DevCluster c = csrc[i]; c.table_off -= base_off; l_clusters[i] = c; }
		if (spec.use_prefix_code) { const int32_t *src = plan.pool_i32 + base_off; int32_t *dst = (int32_t *) l_tab; for (uint32_t i = lane; i < spec.table_span; i += 64) dst[i] = src[i]; }
		else { const uint64_t *src = plan.pool_u64 + base_off; uint64_t *dst = (uint64_t *) l_tab; for (uint32_t i = lane; i < spec.table_span; i += 64) dst[i] = src[i]; }
		t.tree = l_tree; t.cluster_map = (const uint8_t *) l_map; t.clusters = l_clusters; t.alias = (const uint64_t *) l_tab; t.prefix = (const int32_t *) l_tab;
		__syncthreads();
	}
	const DevModFrame f = *plan.frame;
	const uint32_t start_bit = 8u * sec.byte_off + sec.bit_off, end_bit = 8u * (sec.byte_off + sec.size);
	UBits b;
	ub_init<true>(b, (const J40_GLOBAL uint8_t *) plan.codestream, start_bit);
	DevCode code;
	code_init(code, spec, t.clusters, t.cluster_map, t.alias, t.prefix, nullptr);
	const bool prefix = code.use_prefix_code != 0;
	uint32_t state = 0, err = 0;
	int32_t dist_mult = 0;   // LZ77 distance multiplier: widest non-meta channel of this sub-image (j40.h:3841-3844; decode_modular_section)
	if (sec.dist_mult_p1) dist_mult = sec.dist_mult_p1 - 1;
	else for (int32_t cidx = 0; cidx < sec.num_channels; ++cidx) { const ModChan c = mod_channel(plan, sec, cidx); if (!c.meta) dist_mult = mod_max(dist_mult, c.gw); }
	dist_mult = mod_min(dist_mult, 1 << 21);
	int32_t *out = plan.residuals + sec.res_off;
	// n: values written (= the reference's num_decoded); `pend`: the last n - flushed of them, one per lane, not yet in memory
	uint32_t n = 0, flushed = 0;
	int32_t pend = 0;
	auto flush = [&]() {   // (wave-uniform: everybody calls it together)
		const uint32_t k = n - flushed;
		if ((uint32_t) lane < k) out[flushed + (uint32_t) lane] = pend;
		flushed = n;
	};
	auto next_token = [&](const ClusterRegs &cl) { return prefix ? split_prefix_token<true>(b, code.prefix, cl, end_bit, &err) : split_ans_token<true>(b, state, code.alias, code.log_bucket, cl, end_bit, &err); };
	// an LZ77 run in progress: `run_left` values still to copy, value j of the run comes from out[run_base + (run_done + j) % run_dist]
	uint32_t run_left = 0, run_base = 0, run_dist = 0, run_done = 0;
	// A run at distance 1 -- "the value before, n times", what run-length passes of encoders write -- repeats a value this wavefront decoded
	// a moment ago: it is kept (last_val; last_known: false behind a run copied out of memory) and the run becomes stores of it, without the
	// wait for the stores before it and the round trip to the L2 that a run copied out of the window pays (config 1: 24.7 -> 23.6 ms.
	// Also measured in round 6, call O, and not kept: a run's start -- length token, distance symbol, both hybrid integers -- worked out by
	// the lanes of the 64-at-a-time mode like a literal is: 23.6 -> 32 ms, every iteration pays for it and run starts are not where the time is)
	int32_t last_val = 0; bool last_known = false, run_fill = false;
	const bool uses_x = sec.split == 2;
	// Prefix codes, the leaf fixed along a row: SIXTY-FOUR symbols are decoded at once, lane i the one that would start at bit P + i --
	// token, extra bits, value, length, all of it a function of the 33 bits at that position and of the row's cluster -- and the stream's
	// real symbols are then picked out by walking the lengths (two cross-lane reads and an add per symbol instead of a hundred scalar
	// instructions). A lane whose symbol is none of the plain kind (an LZ77 token, a code beyond the first table, an error, more than 32
	// bits in all) says length 255, and the walk hands that one symbol to the scalar decoder below. P: the next unread bit in this mode.
	const bool fast = prefix && !uses_x && !split_no_fast();
	uint32_t P = start_bit, chunk_bit0 = 0xffffffffu;
	__shared__ uint32_t chunk[SPLIT_CHUNK_WORDS + 8];
	const J40_GLOBAL uint32_t *words = (const J40_GLOBAL uint32_t *) plan.codestream;
	for (int32_t cidx = 0; cidx < sec.num_channels && !err; ++cidx) {
		const ModChan chan = mod_channel(plan, sec, cidx);
		const int32_t gw = chan.gw, gh = chan.gh;
		if (gw <= 0 || gh <= 0) continue;
		for (int32_t y = 0; y < gh && !err; ++y) {
			SplitLeaf leaf = split_leaf<true>(t.tree, cidx, sec.sidx, y, 0);
			ClusterRegs cl = load_cluster<true>(code, leaf.ctx);
			const int32_t split_exp = (int32_t) (cl.cfg & 15), hsplit = 1 << split_exp, msb = (int32_t) ((cl.cfg >> 4) & 15), lsb = (int32_t) ((cl.cfg >> 8) & 15), in_token = msb + lsb;
			for (int32_t x = 0; x < gw; ) {
				if (run_left) {
					// the rest of the run that fits this row: the sources lie before the run's start, all of them in memory (flushed there)
					const uint32_t k = mod_min((int32_t) run_left, gw - x);
					if (run_fill) for (uint32_t j0 = 0; j0 < k; j0 += 64) { const uint32_t j = j0 + (uint32_t) lane; if (j < k) out[n + j] = last_val; }
					else for (uint32_t j0 = 0; j0 < k; j0 += 64) {
						const uint32_t j = j0 + (uint32_t) lane;
						// (agent-scope loads: what this wavefront stored a moment ago, from the L2 -- never a line the vector L1 took in earlier; a run
						// that starts before the first decoded value copies zeros, j40.h:2858)
						if (j < k) out[n + j] = run_dist ? __hip_atomic_load(out + run_base + (run_done + j) % run_dist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
					}
					n += k; flushed = n; run_left -= k; run_done += k; x += (int32_t) k;
					continue;
				}
				if (fast) {
					// the stream's next 4 KB wait in LDS: a window's two words are an LDS read away, not a trip to the L2
					if (P < chunk_bit0 || P + 160u > chunk_bit0 + 32u * SPLIT_CHUNK_WORDS) {
						const uint32_t w0 = P >> 5, wlast = (end_bit >> 5) + 6;   // (nothing behind the section's end but the buffer's padding is read)
						for (uint32_t i = (uint32_t) lane; i < SPLIT_CHUNK_WORDS; i += 64) chunk[i] = words[w0 + i < wlast ? w0 + i : wlast];
						chunk_bit0 = w0 << 5;
						__builtin_amdgcn_s_waitcnt(0);
					}
					const uint32_t p = P - chunk_bit0 + (uint32_t) lane, wi = p >> 5, sh = p & 31u;
					const uint64_t ww = (((uint64_t) chunk[wi + 1] << 32) | (uint64_t) chunk[wi]) >> sh;   // >= 33 bits of the stream from bit P + lane on
					const int32_t *table = code.prefix + cl.table_off;
					const int32_t entry = table[(uint32_t) ww & ((1u << cl.fast_len) - 1)];
					const int32_t len1 = entry & 15, tok = entry >> 16;
					const int32_t tk = mod_min(tok, cl.max_token);
					const int32_t midbits = tok < hsplit ? 0 : split_exp - in_token + ((tk - hsplit) >> in_token);
					const int32_t tot = len1 + midbits;
					const bool plain = entry >= 0 && tok < code.min_symbol && tok <= mod_max(cl.max_token, hsplit - 1) && midbits >= 0 && tot <= 32;
					const int32_t mid = (int32_t) ((uint32_t) (ww >> len1) & (midbits >= 32 ? 0xffffffffu : (1u << (midbits & 31)) - 1u));
					const int32_t top = 1 << msb, lo = tk & ((1 << lsb) - 1), hi = (tk >> lsb) & (top - 1);
					const int32_t val = tok < hsplit ? tok : ((top | hi) << ((midbits + lsb) & 31)) | ((mid << lsb) | lo);
					// lanes whose symbol the walk must not take: not of the plain kind, or ending behind the section's end
					const uint64_t stop = __builtin_amdgcn_ballot_w64(!plain || P + (uint32_t) lane + (uint32_t) tot > end_bit);
					const uint32_t rem = (uint32_t) (gw - x);
					uint32_t off = 0, cnt = 0; uint64_t starts = 0; bool slow = false;
					if (!(stop & 1u) && __builtin_amdgcn_readfirstlane(tot) == 0) {
						// a code of ONE symbol that is a literal: no bits, the rest of the row is that value
						const int32_t v = __builtin_amdgcn_readfirstlane(val);
						for (uint32_t j = (uint32_t) lane; j < rem; j += 64) out[n + j] = v;
						n += rem; flushed = n; x = gw;
						last_val = v; last_known = true;
						continue;
					}
					while (off < 64u && cnt < rem) {
						if ((stop >> off) & 1u) { slow = true; break; }
						starts |= (uint64_t) 1 << off;
						off += (uint32_t) __builtin_amdgcn_readlane(tot, (int32_t) off);
						++cnt;
					}
					// the walk's symbols leave in stream order: the k-th start lane stores to n + k
					if ((starts >> lane) & 1u) out[n + (uint32_t) __builtin_popcountll(starts & (((uint64_t) 1 << lane) - 1u))] = val;
					if (cnt) { last_val = __builtin_amdgcn_readlane(val, 63 - __builtin_clzll(starts)); last_known = true; }   // (the walk's last symbol)
					n += cnt; flushed = n; x += (int32_t) cnt; P += off;
					if (!slow) continue;
					{   // this one symbol the long way: the scalar decoder's window set up at P out of the same LDS chunk (three words; the words after
						// them it asks for itself, from memory, ahead of their use)
						const uint32_t rel = P - chunk_bit0, wi = rel >> 5, sh = rel & 31u;
						const uint32_t w0 = uni<true>(chunk[wi]), w1 = uni<true>(chunk[wi + 1]);
						b.bits = (((uint64_t) w1 << 32) | (uint64_t) w0) >> sh; b.nbits = 64 - (int32_t) sh;
						b.pos = ((chunk_bit0 >> 5) + wi + 2) * 4u; b.ahead = chunk[wi + 2];
					}
				}
				if (uses_x) { leaf = split_leaf<true>(t.tree, cidx, sec.sidx, y, x); cl = load_cluster<true>(code, leaf.ctx); }
				int32_t token = next_token(cl);
				if (token < code.min_symbol) {   // (min_symbol is out of reach when the code has no LZ77, j40.h:2823)
					token = split_hybrid<true>(b, token, cl.cfg, cl.max_token, end_bit, &err);
					if (err) break;
					pend = lane == (int32_t) (n - flushed) ? token : pend;   // (lane n - flushed of `pend` := the wave-uniform token: a compare and a select)
					last_val = token; last_known = true;
					++n; ++x;
					if (n - flushed == 64 || fast) flush();   // (the sixty-four-at-a-time mode stores its values itself: nothing may wait in `pend` beside it)
					if (fast) P = ub_position(b);
					continue;
				}
				// an LZ77 run starts (j40.h:2823-2866): its length from this token, its distance from the stream's last context
				const ClusterRegs lz = load_cluster<true>(code, code.num_dist - 1);
				const int32_t num_to_copy = split_hybrid<true>(b, token - code.min_symbol, code.lz_len_cfg, code.lz_len_max_token, end_bit, &err) + code.min_length;
				token = next_token(lz);
				int32_t distance = split_hybrid<true>(b, token, lz.cfg, lz.max_token, end_bit, &err);
				if (err) break;
				if (fast) P = ub_position(b);
				if (!dist_mult) ++distance;
				else if (distance >= 120) distance -= 119;
				else {
					const int32_t special = (int32_t) uni<true>((uint32_t) LZ77_SPECIAL_DISTANCES[distance]);
					distance = ((special >> 4) - 7) + dist_mult * (special & 7);
					if (distance < 1) distance = 1;
				}
				if ((uint32_t) distance > n) distance = (int32_t) n;
				if (distance > (1 << 20)) distance = 1 << 20;
				flush();
				run_fill = distance == 1 && last_known;   // (n >= 1 then: the value before the run is last_val, and stays the last one behind it)
				if (!run_fill) { __builtin_amdgcn_s_waitcnt(0); last_known = false; }   // (the run reads what was just stored; what it ends on is not kept)
				run_left = (uint32_t) mod_max(num_to_copy, 0); run_base = n - (uint32_t) distance; run_dist = (uint32_t) distance; run_done = 0;
				if (run_left > sec.res_count - n) run_left = sec.res_count - n;   // (a run past the section's last sample: the values it still codes are never asked for)
			}
		}
	}
	if (fast && !err) ub_init<true>(b, (const J40_GLOBAL uint8_t *) plan.codestream, P);   // (what follows the last symbol is read through the window again)
	flush();
	split_finish<true>(b, prefix, state, end_bit, f.check_section_end != 0, f.single_declared_end, &err);
	if (lane == 0) { note[0] = err; note[1] = err ? n : 0xffffffffu; note[2] = 0xffffffffu; }
}

// value of `v` in the lane above (lane 0: its own)
__device__ __forceinline__ int32_t split_from_lane_above(int32_t v, int32_t lane) { return __builtin_amdgcn_ds_bpermute(((lane > 0 ? lane - 1 : 0)) << 2, v); }

__global__ void __launch_bounds__(64) k_modular_predict(DevModPlan plan, int32_t first_section, int32_t row_floats) {
	extern __shared__ __attribute__((aligned(16))) int32_t split_rows[];   // [2][row_floats]: the two rows above the band (y0 - 1, y0 - 2)
	const int32_t lane = threadIdx.x;
	const int32_t s = first_section + (int32_t) blockIdx.x, cidx = (int32_t) blockIdx.y;
	const DevModSection sec = plan.sections[s];
	if (!sec.split || sec.preset_status || cidx >= sec.num_channels) return;
	const ModChan chan = mod_channel(plan, sec, cidx);
	const int32_t gw = chan.gw, gh = chan.gh, stride = chan.stride;
	if (gw <= 0 || gh <= 0) return;
	uint32_t chan_base = 0;   // the channel's first ordinal in the section's stream
	for (int32_t k = 0; k < cidx; ++k) { const ModChan c = mod_channel(plan, sec, k); if (c.gw > 0 && c.gh > 0) chan_base += (uint32_t) c.gw * (uint32_t) c.gh; }
	const int32_t *res = plan.residuals + sec.res_off + chan_base;
	const DevTreeNode *tree = plan.tree + sec.tree_off;
	const bool uses_x = sec.split == 2;
	uint32_t povf_at = 0xffffffffu;
	int32_t *above1 = split_rows, *above2 = split_rows + row_floats;
	for (int32_t y0 = 0; y0 < gh; y0 += 64) {
		// the two rows above the band, from the plane (the band before wrote them: this wavefront's own stores, made visible -- and the
		// vector L1 emptied -- by the fence)
		__threadfence();
		__syncthreads();
		for (int32_t x = lane; x < gw; x += 64) {
			above1[x] = y0 > 0 ? (int32_t) chan.base[(size_t) (y0 - 1) * (size_t) stride + (size_t) x] : 0;
			above2[x] = y0 > 1 ? (int32_t) chan.base[(size_t) (y0 - 2) * (size_t) stride + (size_t) x] : 0;
		}
		__syncthreads();
		const int32_t y = y0 + lane;
		const bool row_live = y < gh;
		SplitLeaf leaf = split_leaf<false>(tree, cidx, sec.sidx, row_live ? y : 0, 0);
		int16_t *row = chan.base + (size_t) (row_live ? y : 0) * (size_t) stride;
		const int32_t *rres = res + (size_t) (row_live ? y : 0) * (size_t) gw;
		// the row above as it slides by: values at x - 2 .. x + 2 (nww, nw, n, ne, nee); own row: w, ww; n as it was 1, 2, 3 steps ago
		int32_t r_nww = 0, r_nw = 0, r_n = 0, r_ne = 0, r_nee = 0, c_w = 0, c_ww = 0, n1 = 0, n2 = 0, n3 = 0;
		const int32_t steps = gw + 3 * 63 + 3;
		for (int32_t tstep = 0; tstep < steps; ++tstep) {
			const int32_t x = tstep - 3 * lane - 2;   // (every lane starts two columns early: its window fills with the values at 0, 1, 2)
			// the row above at x + 2: the lane above computed it in the step before (its c_w); lane 0 reads the band's carry-over
			int32_t in_nee = split_from_lane_above(c_w, lane);
			int32_t in_nn = split_from_lane_above(n3, lane);
			if (lane == 0) { in_nee = x + 2 >= 0 && x + 2 < gw ? above1[x + 2] : 0; in_nn = x >= 0 && x < gw ? above2[x] : 0; }
			r_nww = r_nw; r_nw = r_n; r_n = r_ne; r_ne = r_nee; r_nee = in_nee;   // now centred on x
			n3 = n2; n2 = n1; n1 = r_n;   // (n1 = N at x, what the lane below will want as NN three steps from now ... see in_nn)
			if (x >= 0 && x < gw && row_live) {
				ModNeigh p;
				p.w = x > 0 ? c_w : y > 0 ? r_n : 0;
				p.n = y > 0 ? r_n : p.w;
				p.nw = x > 0 && y > 0 ? r_nw : p.w;
				p.ne = x + 1 < gw && y > 0 ? r_ne : p.n;
				p.nn = y > 1 ? in_nn : p.n;
				p.nee = x + 2 < gw && y > 0 ? r_nee : p.ne;
				p.ww = x > 1 ? c_ww : p.w;
				p.nww = x > 1 && y > 0 ? r_nww : p.ww;
				if (uses_x) leaf = split_leaf<false>(tree, cidx, sec.sidx, y, x);
				int32_t v;
				if (!split_sample(rres[x], leaf, p, &v)) { const uint32_t at = chan_base + (uint32_t) y * (uint32_t) gw + (uint32_t) x; povf_at = at < povf_at ? at : povf_at; }
				row[x] = (int16_t) v;
				c_ww = c_w; c_w = v;
			}
		}
	}
	if (povf_at != 0xffffffffu) atomicMin(plan.split_state + 3 * (size_t) s + 2, povf_at);
}

__global__ void k_modular_split_status(DevModPlan plan, int32_t first_section, int32_t num_sections) {
	const int32_t i = (int32_t) (blockIdx.x * blockDim.x + threadIdx.x);
	if (i >= num_sections) return;
	const int32_t s = first_section + i;
	if (!plan.sections[s].split) return;
	const uint32_t *note = plan.split_state + 3 * (size_t) s;
	plan.status[s] = split_status(note[0], note[1], note[2]);
}

void launch_modular_split(const DevModPlan &plan, int32_t first_section, int32_t num_sections, const ModLaunchInfo &info, hipStream_t stream) {
	if (num_sections <= 0 || info.split_sections <= 0) return;
	static const bool once = [] { const char *e = getenv("J40HIP_SPLIT_NO_FAST"); const bool v = e && atoi(e) != 0; if (v) (void) hipMemcpyToSymbol(HIP_SYMBOL(g_split_no_fast), &v, sizeof v); return true; }();
	(void) once;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	const uint32_t lds = align16((uint32_t) info.num_tree_nodes * (uint32_t) sizeof(DevTreeNode)) + align16((uint32_t) info.num_dist + 4)
		+ align16((uint32_t) info.num_clusters * (uint32_t) sizeof(DevCluster)) + align16(info.table_bytes) + 64;
	if (lds <= 150u * 1024u) {   // (beside the kernel's own 4 KB window on the codestream)
		static bool configured = false;
		if (!configured) { (void) hipFuncSetAttribute((const void *) k_modular_tokens<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 154 * 1024); configured = true; }
		hipLaunchKernelGGL(k_modular_tokens<true>, dim3((unsigned) num_sections), dim3(64), lds, stream, plan, first_section);
	} else hipLaunchKernelGGL(k_modular_tokens<false>, dim3((unsigned) num_sections), dim3(64), 0, stream, plan, first_section);
	const int32_t row_floats = (info.split_width + 3) & ~3;
	hipLaunchKernelGGL(k_modular_predict, dim3((unsigned) num_sections, (unsigned) info.split_channels), dim3(64), 2 * (size_t) row_floats * sizeof(int32_t), stream, plan, first_section, row_floats);
	hipLaunchKernelGGL(k_modular_split_status, dim3((unsigned) ((num_sections + 63) / 64)), dim3(64), 0, stream, plan, first_section, num_sections);
}

}  // namespace j40hip
