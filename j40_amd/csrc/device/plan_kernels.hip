// j40_amd/csrc/device/plan_kernels.hip -- the LF-dependent half of a VarDCT frame's plan, built on the device for every frame of
// a batch at once (replaces, in the pipeline, frame.cpp's lf_group_finish -- the reference's
// j40__hf_metadata placement loop, j40.h:6634-6701, and LF index, j40.h:6566-6570 -- and plan_build.cpp's work lists). The
// arithmetic lives in plan_dev.h, which tests/hostsim compiles for the CPU and checks against the host path array by array.
//
//   k_plan_place     one LfGroup per WAVEFRONT: the placement is a serial walk (where a block goes depends on every block before
//                    it) kept to what has to be serial -- DctSelect, first free cell, checks, occupancy --; the varblocks' records
//                    are completed 64 at a time, one per lane (k_plan_place_walk: everything inside the walk, round 3's form;
//                    k_plan_place_lanes: one LfGroup per lane, the first form)
//   k_plan_scan      one lane per frame: where each group's block list and each (DctSelect, LfGroup)'s work items start
//   k_plan_emit      one lane per varblock: its K1 record (block contexts from the LF index of its top-left cell) and K2 record
//   k_plan_verdict   one wavefront per frame, after the entropy kernel: the first failing section in file order
#include <hip/hip_runtime.h>
#include "plan_dev.h"
#include "kernels.h"
#include <algorithm>

namespace j40hip {

// The lane-per-LfGroup form (plan_dev.h's plan_place_lf_group as it stands; J40HIP_PLAN_PLACE_LANES=1): 64 serial walks side by
// side diverge at every branch and every walk waits on its own dependent loads -- 73 ms for the 3072 LfGroups of 256 8K frames.
__global__ void __launch_bounds__(64) k_plan_place_lanes(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nlf) {
	__shared__ uint16_t occ[256 * 64];
	__shared__ uint16_t grp[64 * 64];
	__shared__ uint32_t cls[28 * 64];
	const int32_t i = (int32_t) (blockIdx.x * 64 + threadIdx.x);
	if (i >= nlf) return;
	const DevBatchLf w = lfs[i];
	plan_place_lf_group(builds[w.frame], w.lfg, occ + threadIdx.x, grp + threadIdx.x, cls + threadIdx.x, 64);
}

// One LfGroup per WAVEFRONT, the same walk with all of its state in registers and every decision wave-uniform (scalar unit):
//   * lane c of four registers holds the occupancy of columns c, c + 64, c + 128, c + 192 (the row below the lowest cell a placed
//     block occupies there); one ballot per quarter gives a row's free cells as a bit mask, and the walk jumps from free cell to
//     free cell with a find-first-set -- occupied cells cost nothing. A block never straddles a group (32 cells), hence never a
//     quarter. Blocks one cell high (the 8x8 transforms: most of them) only clear bits of the row's mask;
//   * the two rows of the varblock-info channel are read 64 entries at a time, one per lane (coalesced), an entry is a readlane;
//   * lane g counts the blocks of group g, lane d those of DctSelect d; the thresholds of the quantisation-field index sit one per
//     lane and are counted with a ballot; the transforms' sizes sit one per lane;
//   * the records of 64 consecutive varblocks collect one per lane and leave as one coalesced store.
// Same results as plan_place_lf_group (which tests/hostsim checks against the host path): tests/test_device_stages.py runs every form.
__device__ __forceinline__ int32_t pp_rl(int32_t v, int32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int32_t pp_sc(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ void __launch_bounds__(64) k_plan_place_walk(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nlf) {
	const int32_t lane = threadIdx.x;
	const DevBatchLf w = lfs[blockIdx.x];
	const DevPlanBuild &pb = builds[w.frame];
	const int32_t g = w.lfg;
	DevLfGroup *ggp = pb.lf_groups + g;
	DevLfSlot *slot = pb.lf_slots + g;
	const int32_t w8 = pp_sc(ggp->width8), h8 = pp_sc(ggp->height8), cell_base = pp_sc(ggp->cell_base), vb_base = pp_sc(ggp->vb_base);
	const int32_t ggx = g % pb.ggcolumns, ggy = g / pb.ggcolumns;
	uint32_t err = (uint32_t) pp_sc((int32_t) slot->status), used = 0;
	const int32_t nbv = pp_sc(slot->nb_varblocks), nb_qf_thr = pp_sc(pb.nb_qf_thr);
	const int32_t my_thr = lane < nb_qf_thr ? pb.qf_thr[lane < 15 ? lane : 14] : 0;
	const int32_t my_dims = lane < 27 ? (int32_t) DEV_DCT_SELECT[lane][0] | ((int32_t) DEV_DCT_SELECT[lane][1] << 8) : 0;
	int32_t occ0 = 0, occ1 = 0, occ2 = 0, occ3 = 0;
	int32_t grp_cnt = 0, cls_cnt = 0;   // lane = group inside the LfGroup / DctSelect
	int32_t voff = 0, coeffoff = 0;
	if (!err) {
		const int16_t *info0 = pb.vbinfo + 2 * (size_t) cell_base, *info1 = info0 + nbv;
		DevVbRec *recs = pb.vb_recs + vb_base;
		const int32_t coeff_limit = w8 * h8 * 64;
		int32_t chunk = -1, i0 = 0, i1 = 0;   // info entries chunk * 64 + lane
		uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;   // the record of varblock (voff & ~63) + lane
		for (int32_t y0 = 0; y0 < h8 && !err; ++y0) {
			for (int32_t q = 0; q < 4 && !err; ++q) {
				if (q * 64 >= w8) break;
				const int32_t occq = q == 0 ? occ0 : q == 1 ? occ1 : q == 2 ? occ2 : occ3;
				uint64_t free_mask = __builtin_amdgcn_ballot_w64(occq <= y0 && q * 64 + lane < w8);
				while (free_mask) {
					const int32_t bit = (int32_t) __builtin_ctzll(free_mask), x0 = q * 64 + bit;
					if (voff >= nbv) { err = ERR_VBLK; break; }
					if ((voff >> 6) != chunk) {
						chunk = voff >> 6;
						const int32_t k = chunk * 64 + lane;
						i0 = k < nbv ? (int32_t) info0[k] : 0; i1 = k < nbv ? (int32_t) info1[k] : 0;
					}
					const int32_t dctsel = pp_rl(i0, voff & 63), hfmul_m1 = pp_rl(i1, voff & 63);
					if (dctsel < 0 || dctsel >= 27) { err = ERR_DCTQ; break; }
					const int32_t dims = pp_rl(my_dims, dctsel), log_rows = dims & 255, log_columns = dims >> 8;
					const int32_t vw8 = 1 << (log_columns - 3), vh8 = 1 << (log_rows - 3), x1 = x0 + vw8 - 1, y1 = y0 + vh8 - 1;
					if (!(x1 < w8 && (x0 >> PLAN_LOG_GSIZE8) == (x1 >> PLAN_LOG_GSIZE8)) || !(y1 < h8 && (y0 >> PLAN_LOG_GSIZE8) == (y1 >> PLAN_LOG_GSIZE8))) { err = ERR_VBLK; break; }
					if (coeffoff + (1 << (log_rows + log_columns)) > coeff_limit) { err = ERR_VBLK; break; }
					free_mask &= ~((((uint64_t) 1 << vw8) - 1) << bit);   // (vw8 <= 32)
					if (vh8 > 1) {
						const bool mine = lane >= bit && lane < bit + vw8;
						const int32_t below = y1 + 1;
						if (q == 0) occ0 = mine && occ0 < below ? below : occ0; else if (q == 1) occ1 = mine && occ1 < below ? below : occ1;
						else if (q == 2) occ2 = mine && occ2 < below ? below : occ2; else occ3 = mine && occ3 < below ? below : occ3;
					}
					const int32_t qf = (int32_t) __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < nb_qf_thr && hfmul_m1 >= my_thr));
					const int32_t grp = (y0 >> PLAN_LOG_GSIZE8) * 8 + (x0 >> PLAN_LOG_GSIZE8);
					const int32_t rank_g = pp_rl(grp_cnt, grp), rank_c = pp_rl(cls_cnt, dctsel);
					grp_cnt += lane == grp; cls_cnt += lane == dctsel;
					if (lane == (voff & 63)) {
						r0 = (uint32_t) (coeffoff + qf);
						r1 = (uint32_t) (uint16_t) (int16_t) hfmul_m1 | (uint32_t) x0 << 16 | (uint32_t) y0 << 24;
						r2 = (uint32_t) dctsel | (uint32_t) grp << 8 | (uint32_t) rank_g << 16;
						r3 = (uint32_t) rank_c;
					}
					used |= 1u << dctsel;
					coeffoff += 1 << (log_rows + log_columns);
					++voff;
					if ((voff & 63) == 0) ((uint4 *) recs)[voff - 64 + lane] = make_uint4(r0, r1, r2, r3);
				}
			}
		}
		if ((voff & 63) != 0 && lane < (voff & 63)) ((uint4 *) recs)[(voff & ~63) + lane] = make_uint4(r0, r1, r2, r3);   // (also after an error: the blocks placed before it)
		if (!err && voff != nbv) err = ERR_VBLK;
	}
	if (lane == 0) { slot->status = err; slot->placed = voff; slot->dct_used = used; ggp->nb_varblocks = voff; }
	{   // every group lies in exactly one LfGroup: plain stores (lane = gy * 8 + gx)
		const int32_t gx = lane & 7, gy = lane >> 3;
		if ((gx << PLAN_LOG_GSIZE8) < w8 && (gy << PLAN_LOG_GSIZE8) < h8) pb.group_count[(ggy * 8 + gy) * pb.gcolumns + ggx * 8 + gx] = (uint32_t) grp_cnt;
	}
	if (lane < 28) pb.class_count[g * 28 + lane] = lane < 27 ? (uint32_t) cls_cnt : 0u;
}

// The form above does everything of a varblock inside the walk: 1800 cycles per varblock, 24 ms for the 3072 LfGroups of 256 8K frames --
// two thirds of the batch's plan stage, on the critical path in front of the entropy decode. What the walk NEEDS per varblock is
// little: its DctSelect (-> its size), the first free cell, the checks, the free mask and the occupancy. Everything else -- the
// coefficient offset (a prefix sum of the sizes), the quantisation-field index, the group, the ranks within group and class, the
// record -- is a function of (x0, y0, DctSelect, HfMul) and of counts, and is computed for 64 varblocks at once, one per lane, each
// time the walk has placed 64: the walk leaves (x0, y0) in the varblock's lane (a compare and a select), its DctSelect and HfMul are there
// already (the info channel's entries sit one per lane). The ranks: the count before the chunk (lane = group / class, fetched with a
// bpermute) plus the varblocks of the same group / class in lower lanes -- one ballot per distinct group / class of the chunk.
// Same products as the walk above and as plan_place_lf_group (tests/test_device_stages.py pins them against the reference's
// internals -- the earlier forms too, each in a process of its own); J40HIP_PLAN_PLACE_FORM=1 selects the walk above.
__device__ __forceinline__ int32_t pp_below(uint64_t m) { return (int32_t) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)); }   // set bits of m below this lane

__global__ void __launch_bounds__(64) k_plan_place(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nlf) {
	const int32_t lane = threadIdx.x;
	const DevBatchLf w = lfs[blockIdx.x];
	const DevPlanBuild &pb = builds[w.frame];
	const int32_t g = w.lfg;
	DevLfGroup *ggp = pb.lf_groups + g;
	DevLfSlot *slot = pb.lf_slots + g;
	const int32_t w8 = pp_sc(ggp->width8), h8 = pp_sc(ggp->height8), cell_base = pp_sc(ggp->cell_base), vb_base = pp_sc(ggp->vb_base);
	const int32_t ggx = g % pb.ggcolumns, ggy = g / pb.ggcolumns;
	uint32_t err = (uint32_t) pp_sc((int32_t) slot->status), used = 0;
	const int32_t nbv = pp_sc(slot->nb_varblocks), nb_qf_thr = pp_sc(pb.nb_qf_thr);
	const int32_t my_thr = lane < nb_qf_thr ? pb.qf_thr[lane < 15 ? lane : 14] : 0;
	const int32_t my_dims = lane < 27 ? (int32_t) DEV_DCT_SELECT[lane][0] | ((int32_t) DEV_DCT_SELECT[lane][1] << 8) : 0;
	int32_t occ0 = 0, occ1 = 0, occ2 = 0, occ3 = 0;
	int32_t grp_cnt = 0, cls_cnt = 0;   // lane = group inside the LfGroup / DctSelect
	int32_t voff = 0;
	if (!err) {
		const int16_t *info0 = pb.vbinfo + 2 * (size_t) cell_base, *info1 = info0 + nbv;
		DevVbRec *recs = pb.vb_recs + vb_base;
		const int32_t coeff_limit = w8 * h8 * 64;
		int32_t coeffoff = 0;              // the walk's own sum (its check against coeff_limit)
		int32_t chunk_coeffoff = 0;        // the coefficient offset of the first varblock that has no record yet
		int32_t chunk = -1, i0 = 0, i1 = 0;   // info entries chunk * 64 + lane
		int32_t pos = 0;                   // x0 | y0 << 8 of varblock (voff & ~63) + lane
		// the records of varblocks first .. first + n - 1 (n <= 64, their data in lanes 0 .. n - 1)
		auto records = [&](int32_t first, int32_t n) {
			const bool valid = lane < n;
			const int32_t dctsel = valid ? i0 : 0, hfmul_m1 = i1;
			const int32_t dims = __builtin_amdgcn_ds_bpermute(dctsel << 2, my_dims);
			const int32_t size = valid ? 1 << ((dims & 255) + (dims >> 8)) : 0;
			int32_t upto = size;   // inclusive prefix sum
			for (int32_t d = 1; d < 64; d <<= 1) { const int32_t o = __shfl_up(upto, d); upto += lane >= d ? o : 0; }
			const int32_t my_off = chunk_coeffoff + upto - size;
			chunk_coeffoff += pp_rl(upto, 63);
			int32_t qf = 0;
			for (int32_t t = 0; t < nb_qf_thr; ++t) qf += hfmul_m1 >= pp_rl(my_thr, t) ? 1 : 0;
			const int32_t x0 = pos & 255, y0 = (pos >> 8) & 255;
			const int32_t grp = (y0 >> PLAN_LOG_GSIZE8) * 8 + (x0 >> PLAN_LOG_GSIZE8);
			int32_t rank_g = __builtin_amdgcn_ds_bpermute(grp << 2, grp_cnt), rank_c = __builtin_amdgcn_ds_bpermute(dctsel << 2, cls_cnt);
			for (uint64_t todo = __builtin_amdgcn_ballot_w64(valid); todo; ) {
				const int32_t key = pp_rl(grp, (int32_t) __builtin_ctzll(todo));
				const uint64_t same = __builtin_amdgcn_ballot_w64(valid && grp == key);
				rank_g += valid && grp == key ? pp_below(same) : 0;
				grp_cnt += lane == key ? (int32_t) __builtin_popcountll(same) : 0;
				todo &= ~same;
			}
			for (uint64_t todo = __builtin_amdgcn_ballot_w64(valid); todo; ) {
				const int32_t key = pp_rl(dctsel, (int32_t) __builtin_ctzll(todo));
				const uint64_t same = __builtin_amdgcn_ballot_w64(valid && dctsel == key);
				rank_c += valid && dctsel == key ? pp_below(same) : 0;
				cls_cnt += lane == key ? (int32_t) __builtin_popcountll(same) : 0;
				todo &= ~same;
			}
			if (valid) ((uint4 *) recs)[first + lane] = make_uint4((uint32_t) (my_off + qf), (uint32_t) (uint16_t) (int16_t) hfmul_m1 | (uint32_t) x0 << 16 | (uint32_t) y0 << 24,
				(uint32_t) dctsel | (uint32_t) grp << 8 | (uint32_t) rank_g << 16, (uint32_t) rank_c);
		};
		for (int32_t y0 = 0; y0 < h8 && !err; ++y0) {
			for (int32_t q = 0; q < 4 && !err; ++q) {
				if (q * 64 >= w8) break;
				const int32_t occq = q == 0 ? occ0 : q == 1 ? occ1 : q == 2 ? occ2 : occ3;
				uint64_t free_mask = __builtin_amdgcn_ballot_w64(occq <= y0 && q * 64 + lane < w8);
				while (free_mask) {
					const int32_t bit = (int32_t) __builtin_ctzll(free_mask), x0 = q * 64 + bit;
					if (voff >= nbv) { err = ERR_VBLK; break; }
					if ((voff >> 6) != chunk) {
						chunk = voff >> 6;
						const int32_t k = chunk * 64 + lane;
						i0 = k < nbv ? (int32_t) info0[k] : 0; i1 = k < nbv ? (int32_t) info1[k] : 0;
					}
					const int32_t dctsel = pp_rl(i0, voff & 63);
					if (dctsel < 0 || dctsel >= 27) { err = ERR_DCTQ; break; }
					const int32_t dims = pp_rl(my_dims, dctsel), log_rows = dims & 255, log_columns = dims >> 8;
					const int32_t vw8 = 1 << (log_columns - 3), vh8 = 1 << (log_rows - 3), x1 = x0 + vw8 - 1, y1 = y0 + vh8 - 1;
					if (!(x1 < w8 && (x0 >> PLAN_LOG_GSIZE8) == (x1 >> PLAN_LOG_GSIZE8)) || !(y1 < h8 && (y0 >> PLAN_LOG_GSIZE8) == (y1 >> PLAN_LOG_GSIZE8))) { err = ERR_VBLK; break; }
					if (coeffoff + (1 << (log_rows + log_columns)) > coeff_limit) { err = ERR_VBLK; break; }
					free_mask &= ~((((uint64_t) 1 << vw8) - 1) << bit);   // (vw8 <= 32)
					if (vh8 > 1) {
						const bool mine = lane >= bit && lane < bit + vw8;
						const int32_t below = y1 + 1;
						if (q == 0) occ0 = mine && occ0 < below ? below : occ0; else if (q == 1) occ1 = mine && occ1 < below ? below : occ1;
						else if (q == 2) occ2 = mine && occ2 < below ? below : occ2; else occ3 = mine && occ3 < below ? below : occ3;
					}
					pos = lane == (voff & 63) ? x0 | y0 << 8 : pos;
					used |= 1u << dctsel;
					coeffoff += 1 << (log_rows + log_columns);
					++voff;
					if ((voff & 63) == 0) records(voff - 64, 64);
				}
			}
		}
		if ((voff & 63) != 0) records(voff & ~63, voff & 63);   // (also after an error: the blocks placed before it)
		if (!err && voff != nbv) err = ERR_VBLK;
	}
	if (lane == 0) { slot->status = err; slot->placed = voff; slot->dct_used = used; ggp->nb_varblocks = voff; }
	{   // every group lies in exactly one LfGroup: plain stores (lane = gy * 8 + gx)
		const int32_t gx = lane & 7, gy = lane >> 3;
		if ((gx << PLAN_LOG_GSIZE8) < w8 && (gy << PLAN_LOG_GSIZE8) < h8) pb.group_count[(ggy * 8 + gy) * pb.gcolumns + ggx * 8 + gx] = (uint32_t) grp_cnt;
	}
	if (lane < 28) pb.class_count[g * 28 + lane] = lane < 27 ? (uint32_t) cls_cnt : 0u;
}

__global__ void __launch_bounds__(64) k_plan_scan(const DevPlanBuild *builds, int32_t nframes) {
	const int32_t f = (int32_t) (blockIdx.x * 64 + threadIdx.x);
	if (f < nframes) plan_scan_frame(builds[f]);
}

__global__ void __launch_bounds__(256) k_plan_emit(const DevPlanBuild *builds, const DevBatchLf *lfs) {
	const DevBatchLf w = lfs[blockIdx.y];
	const DevPlanBuild &pb = builds[w.frame];
	const int32_t v = (int32_t) (blockIdx.x * 256 + threadIdx.x);
	if (v < pb.lf_slots[w.lfg].placed) plan_emit_varblock(pb, w.lfg, v);
}

__global__ void __launch_bounds__(64) k_plan_verdict(const DevPlanBuild *builds, const DevPlan *plans) {
	const DevPlanBuild &pb = builds[blockIdx.x];
	const DevPlan &plan = plans[blockIdx.x];
	const int32_t lane = threadIdx.x, nsec = plan.frame->num_passes * pb.num_groups;
	uint64_t best = ~(uint64_t) 0;
	uint32_t flags = 0, used = 0;
	for (int32_t g = lane; g < pb.num_lf_groups; g += 64) {
		const DevLfSlot sl = pb.lf_slots[g];
		if (sl.status == (uint32_t) ERR_LFFB) flags |= 1u;
		used |= sl.dct_used;
		const uint64_t k = plan_verdict_key(sl.status, pb.lf_section_off[g]);
		best = k < best ? k : best;
	}
	for (int32_t i = lane; i < nsec; i += 64) {
		const uint32_t st = plan.status[i];
		if (st == (uint32_t) ERR_EVOF) flags |= 2u;
		const uint64_t k = plan_verdict_key(st, plan.sections[i].byte_off);
		best = k < best ? k : best;
	}
	for (int d = 32; d >= 1; d >>= 1) {
		const uint64_t o = __shfl_xor(best, d);
		best = o < best ? o : best;
		flags |= __shfl_xor(flags, d); used |= __shfl_xor(used, d);
	}
	if (lane == 0) { pb.verdict[0] = best == ~(uint64_t) 0 ? 0u : (uint32_t) best; pb.verdict[1] = flags; pb.verdict[2] = used; pb.verdict[3] = (uint32_t) pb.class_start[27]; }
}

// clears the per-block event tables (DevPlan::block_events, 16 bytes per cell) of every sparse frame of a batch
__global__ void __launch_bounds__(256) k_clear_block_events(const DevPlan *plans, const DevPlanBuild *builds) {
	const DevPlan &plan = plans[blockIdx.y];
	if (!plan.block_events) return;
	const uint32_t cells = builds[blockIdx.y].cells;
	uint4 *p = (uint4 *) plan.block_events;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < cells; i += gridDim.x * 256u) p[i] = make_uint4(0, 0, 0, 0);
}
void launch_clear_block_events(const DevPlan *plans, const DevPlanBuild *builds, int32_t nframes, size_t max_frame_cells, hipStream_t stream) {
	if (nframes <= 0 || !max_frame_cells) return;
	const unsigned gx = (unsigned) std::min<size_t>((max_frame_cells + 255) / 256, 256);
	hipLaunchKernelGGL(k_clear_block_events, dim3(gx, (unsigned) nframes), dim3(256), 0, stream, plans, builds);
}

void launch_plan_build(const DevPlanBuild *builds, const DevBatchLf *lfs, int32_t nframes, int32_t nlf, int32_t max_lf_cells, hipStream_t stream) {
	if (nframes <= 0 || nlf <= 0) return;
	static const bool lanes_form = [] { const char *e = getenv("J40HIP_PLAN_PLACE_LANES"); return e && atoi(e); }();
	static const int form = [] { const char *e = getenv("J40HIP_PLAN_PLACE_FORM"); return e ? atoi(e) : 0; }();
	if (lanes_form) hipLaunchKernelGGL(k_plan_place_lanes, dim3((unsigned) ((nlf + 63) / 64)), dim3(64), 0, stream, builds, lfs, nlf);
	else if (form == 1) hipLaunchKernelGGL(k_plan_place_walk, dim3((unsigned) nlf), dim3(64), 0, stream, builds, lfs, nlf);
	else hipLaunchKernelGGL(k_plan_place, dim3((unsigned) nlf), dim3(64), 0, stream, builds, lfs, nlf);
	hipLaunchKernelGGL(k_plan_scan, dim3((unsigned) ((nframes + 63) / 64)), dim3(64), 0, stream, builds, nframes);
	hipLaunchKernelGGL(k_plan_emit, dim3((unsigned) ((max_lf_cells + 255) / 256), (unsigned) nlf), dim3(256), 0, stream, builds, lfs);
}

void launch_plan_verdict(const DevPlanBuild *builds, const DevPlan *plans, int32_t nframes, hipStream_t stream) {
	if (nframes > 0) hipLaunchKernelGGL(k_plan_verdict, dim3((unsigned) nframes), dim3(64), 0, stream, builds, plans);
}

} // namespace j40hip
