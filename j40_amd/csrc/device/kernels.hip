// j40_amd/csrc/device/kernels.hip -- the HIP kernels of the VarDCT hot path for gfx950.
//
//   k_hf_entropy        K1: one sequential rANS/prefix stream per lane, one 256x256 group per lane
//                       (replaces j40__pass_group -> j40__hf_coeffs, j40.h:7007 / 6888)
//   k_vardct_dct<R,C>   K2: dequantise + chroma-from-luma + 2-D inverse DCT in LDS + XYB->sRGB + pack,
//                       NB varblocks per workgroup (replaces j40__dequant_hf, j40.h:7053,
//                       j40__combine_vardct_from_lf_group, j40.h:7099, j40__render_to_u8x4_rgba, j40.h:7910)
//   k_vardct_special    K2s: the 8x8 "special" transforms (Hornuss, DCT2x2, DCT4x4, DCT4x8/8x4, AFV)
//   k_vardct_large      K2l: 128/256-sized transforms (large_dev.h): the top levels of the recursion over the tile in LDS, 64-point
//                       sub-vectors in registers; tiles up to 128x128 live in LDS, 256-sized ones in an HBM scratch
//
// Compiled with -ffp-contract=off: the float path has to keep the reference's operation order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <type_traits>
#include <cstring>
#include <cstdlib>
#include "hf_dev.h"
#include "hf_lanes_dev.h"
#include "idct_dev.h"
#include "vardct_dev.h"
#include "special8_dev.h"
#include "large_dev.h"
#include "k2_iter_dev.h"
#include "hf_uni_dev.h"
#include "restore_dev.h"
#include "kernels.h"

namespace j40hip {

__constant__ float c_half_secants[256];
__constant__ float c_afv_basis[256];
__device__ float c_srgb_thr[SRGB_TABLE_FLOATS];

// the pixel kernels keep the sRGB threshold table (idct_dev.h) in LDS; 8-bit frames only
#define J40_STAGE_SRGB_THRESHOLDS(f) \
	__shared__ float s_srgb_thr[SRGB_TABLE_FLOATS]; \
	for (int32_t i_ = threadIdx.x; i_ < SRGB_TABLE_FLOATS; i_ += blockDim.x) s_srgb_thr[i_] = c_srgb_thr[i_]; \
	const J40_LDS float *srgb_thr = (const J40_LDS float *) s_srgb_thr

void upload_constant_tables(const float *half_secants, const float *afv_basis, const float *srgb_thr, hipStream_t stream) {
	(void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_srgb_thr), srgb_thr, sizeof(float) * SRGB_TABLE_FLOATS, 0, hipMemcpyHostToDevice, stream);
	(void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_half_secants), half_secants, sizeof(float) * 256, 0, hipMemcpyHostToDevice, stream);
	(void) hipMemcpyToSymbolAsync(HIP_SYMBOL(c_afv_basis), afv_basis, sizeof(float) * 256, 0, hipMemcpyHostToDevice, stream);
}

// ------------------------------------------------------------------------------------------------
// K1

// One wavefront per group, HF_WAVES groups per workgroup. Lane 0 of each wave runs the sequential
// decoder; all lanes of the workgroup first stage what the decoder keeps touching into LDS:
//   shared by the workgroup: context -> cluster map, cluster descriptors, rANS alias / prefix tables
//                            of the current pass, block context map, the two small context tables
//   per wave:                the group's block list (visiting order) and its non-zero-count scratch
// so the serial path waits on LDS (~64 cycles) instead of HBM/L2 (500-900 cycles) for every symbol.
// The bitstream itself is read through a one-word-ahead prefetch (entropy_dev.h).
struct HfLdsLayout {
	uint32_t off_bctx, off_nnz, off_freq, off_map, off_clusters, off_tables, off_wave;  // byte offsets
	uint32_t wave_bytes, off_wave_blocks;  // per-wave area: nonzeros first, then the block list
	uint32_t total;
};

template <bool TABLES_IN_LDS>
__global__ void __launch_bounds__(64 * HF_WAVES) k_hf_entropy(DevPlan plan, int32_t first_group, int32_t num_groups, HfLdsLayout lay) {
	extern __shared__ __attribute__((aligned(16))) uint8_t hf_lds[];
	const DevFrame &f = *plan.frame;
	const int32_t tid = threadIdx.x, lane = tid & 63;
	const int32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction
	const int32_t local = blockIdx.x * HF_WAVES + wave;
	const bool active = local < num_groups;
	const int32_t g = first_group + local;

	uint8_t *l_bctx = hf_lds + lay.off_bctx;
	int16_t *l_nnz = (int16_t *) (hf_lds + lay.off_nnz);
	int8_t *l_freq = (int8_t *) (hf_lds + lay.off_freq);
	uint8_t *l_map = hf_lds + lay.off_map;
	DevCluster *l_clusters = (DevCluster *) (hf_lds + lay.off_clusters);
	uint8_t *l_tables = hf_lds + lay.off_tables;
	int8_t *l_nonzeros = (int8_t *) (hf_lds + lay.off_wave + wave * lay.wave_bytes);
	DevGroupBlock *l_blocks = (DevGroupBlock *) (hf_lds + lay.off_wave + wave * lay.wave_bytes + lay.off_wave_blocks);

	// frame-level tables
	{
		const uint8_t *src = plan.pool_u8 + plan.block_ctx_map_off;
		const int32_t n = 39 * f.lfidx_size * (f.nb_qf_thr + 1);
		for (int32_t i = tid; i < n; i += blockDim.x) l_bctx[i] = src[i];
		if (tid < 64) { l_nnz[tid] = DEV_NNZ_CTX2[tid]; l_freq[tid] = DEV_FREQ_CTX2[tid]; }
	}
	HfTables t;
	t.block_ctx_map = l_bctx; t.nnz_ctx2 = l_nnz; t.freq_ctx2 = l_freq;
	t.nonzeros = l_nonzeros;
	t.window = plan.lz_window && active ? plan.lz_window + (size_t) g * plan.lz_window_size : nullptr;
	t.nblocks = 0; t.blocks = l_blocks;
	t.block_first = 0; t.ev_first = t.ev_end = 0;
	if (active) {
		const uint32_t b0 = plan.group_block_start[g], b1 = plan.group_block_start[g + 1];
		t.block_first = b0;
		if (f.sparse_coeffs) { t.ev_first = plan.ev_range[2 * g]; t.ev_end = plan.ev_range[2 * g + 1]; }
		t.nblocks = (int32_t) (b1 - b0);
		const uint64_t *src = (const uint64_t *) (plan.group_blocks + b0);   // 8-byte entries
		uint64_t *dst = (uint64_t *) l_blocks;
		for (int32_t i = lane; i < t.nblocks; i += 64) dst[i] = src[i];
	}
	for (int32_t pass = 0; pass < f.num_passes; ++pass) {
		const DevCodeSpec &spec = plan.coeff_specs[pass];
		if (TABLES_IN_LDS) {
			__syncthreads();  // everyone is done with the previous pass' tables
			const uint8_t *msrc = plan.pool_u8 + spec.cluster_map_off;
			for (int32_t i = tid; i < spec.num_dist; i += blockDim.x) l_map[i] = msrc[i];
			// tables of this spec's clusters are contiguous in their pool: copy the span, rebase offsets
			const DevCluster *csrc = plan.clusters + spec.cluster_off;
			const uint32_t base_off = csrc[0].table_off;
			for (int32_t i = tid; i < spec.num_clusters; i += blockDim.x) { DevCluster c = csrc[i]; c.table_off -= base_off; l_clusters[i] = c; }
			if (spec.use_prefix_code) {
				const int32_t *src = plan.pool_i32 + base_off; int32_t *dst = (int32_t *) l_tables;
				for (uint32_t i = tid; i < spec.table_span; i += blockDim.x) dst[i] = src[i];
			} else {
				const uint64_t *src = plan.pool_u64 + base_off; uint64_t *dst = (uint64_t *) l_tables;
				for (uint32_t i = tid; i < spec.table_span; i += blockDim.x) dst[i] = src[i];
			}
			t.clusters = l_clusters; t.cluster_map = l_map; t.alias = (const uint64_t *) l_tables; t.prefix = (const int32_t *) l_tables;
		} else {
			t.clusters = plan.clusters + spec.cluster_off; t.cluster_map = plan.pool_u8 + spec.cluster_map_off;
			t.alias = plan.pool_u64; t.prefix = plan.pool_i32;
		}
		__syncthreads();
		if (active) {
			// every lane of the wave runs the same (scalarised) decoder on the same section; duplicate
			// stores hit the same addresses with the same values
			const DevSection &sec = plan.sections[pass * f.num_groups + g];
			const uint32_t err = f.sparse_coeffs ? decode_hf_section<true, true>(plan, f, spec, t, pass, sec) : decode_hf_section<false, true>(plan, f, spec, t, pass, sec);
			if (lane == 0) plan.status[pass * f.num_groups + g] = err;
		}
	}
}

// K1, latency form, fast path (hf_uni_dev.h): single-pass frames coded with rANS and no LZ77, at most 64 clusters, tables that fit
// in LDS -- what k_hf_lanes takes in a batch. One section per wavefront like k_hf_entropy, the same packed tables as k_hf_lanes.
// `order`: null, or the groups in the order the launch takes them (the single-image path's two launches: runtime.hip, "two phases").
__global__ void __launch_bounds__(64 * HF_WAVES) k_hf_entropy_fast(DevPlan plan, int32_t first_group, int32_t num_groups, uint32_t tables_bytes, uint32_t wave_bytes, const uint32_t *order) {
	extern __shared__ __attribute__((aligned(16))) uint8_t hf_lds[];
	const DevFrame &f = *plan.frame;
	const int32_t tid = threadIdx.x, lane = tid & 63;
	const int32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int32_t local = blockIdx.x * HF_WAVES + wave;
	const bool active = local < num_groups;
	const int32_t g = order ? (active ? (int32_t) __builtin_amdgcn_readfirstlane((int32_t) order[first_group + local]) : 0) : first_group + local;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	const DevCodeSpec &spec = plan.coeff_specs[0];
	const int32_t num_dist = spec.num_dist, num_clusters = spec.num_clusters, log_alpha = spec.log_alpha_size;
	J40_LDS uint8_t *lds = (J40_LDS uint8_t *) hf_lds;
	J40_LDS uint8_t *l_map = lds;
	J40_LDS uint64_t *l_alias = (J40_LDS uint64_t *) (lds + align16((uint32_t) num_dist));
	{
		const uint32_t *msrc = (const uint32_t *) (plan.pool_u8 + spec.cluster_map_off);   // the u8 pool keeps 4-byte alignment per table
		J40_LDS uint32_t *mdst = (J40_LDS uint32_t *) l_map;
		for (int32_t i = tid; i < (num_dist + 3) / 4; i += blockDim.x) mdst[i] = msrc[i];
		const uint64_t *asrc = plan.pool_u64 + plan.clusters[spec.cluster_off].table_off;
		for (uint32_t i = tid; i < ((uint32_t) num_clusters << log_alpha); i += blockDim.x) l_alias[i] = asrc[i];
	}
	UniTables t;
	t.ctx_map = l_map; t.alias = l_alias; t.log_alpha = log_alpha; t.log_bucket = 12 - log_alpha; t.num_dist = num_dist;
	{
		const int32_t *csrc = plan.pool_i32 + spec.lane_cfg_off;
		lr_fill(t.cfg, [&](int32_t i) { return i < num_clusters ? csrc[i] : 0; });
		lr_fill(t.nnz2, [&](int32_t i) { return (int32_t) DEV_NNZ_CTX2[i]; });
		lr_fill(t.freq2, [&](int32_t i) { return (int32_t) DEV_FREQ_CTX2[i]; });
		lr_fill(t.dct, [&](int32_t i) { return i < 27 ? (int32_t) DEV_DCT_SELECT[i][0] | ((int32_t) DEV_DCT_SELECT[i][1] << 8) | ((int32_t) DEV_DCT_SELECT[i][2] << 16) : 0; });
	}
	HfTables h;
	h.clusters = nullptr; h.cluster_map = nullptr; h.alias = nullptr; h.prefix = nullptr; h.block_ctx_map = nullptr; h.nnz_ctx2 = nullptr; h.freq_ctx2 = nullptr; h.window = nullptr;
	h.nonzeros = (int8_t *) (hf_lds + tables_bytes + (uint32_t) wave * wave_bytes);
	DevGroupBlock *l_blocks = (DevGroupBlock *) (hf_lds + tables_bytes + (uint32_t) wave * wave_bytes + 32 * 32 * 3);
	h.blocks = l_blocks; h.nblocks = 0; h.block_first = 0; h.ev_first = h.ev_end = 0;
	if (active) {
		const uint32_t b0 = plan.group_block_start[g], b1 = plan.group_block_start[g + 1];
		h.block_first = b0; h.ev_first = plan.ev_range[2 * g]; h.ev_end = plan.ev_range[2 * g + 1];
		h.nblocks = (int32_t) (b1 - b0);
		const uint64_t *src = (const uint64_t *) (plan.group_blocks + b0);
		uint64_t *dst = (uint64_t *) l_blocks;
		for (int32_t i = lane; i < h.nblocks; i += 64) dst[i] = src[i];
	}
	__syncthreads();
	if (active) {
		const uint32_t err = decode_hf_section_fast<true>(plan, f, t, h, plan.sections[g]);
		if (lane == 0) plan.status[g] = err;
	}
}

// K1, throughput form: one section per LANE, 64 groups of one frame per wavefront, one wavefront per
// workgroup, any number of frames per launch (`work` lists the (frame, first group, count) chunks).
// The lanes run decode_hf_section_flat: every iteration all 64 decode one symbol together. Tables of
// the chunk's frame are staged in LDS and read with per-lane addresses; block lists, the non-zero
// scratch and the bitstreams stay in HBM/L2 (per-lane, touched once per block / once per 32 bits).
// A lone wave is slower per section than the scalarised k_hf_entropy, but a batch fills every SIMD
// with several such waves: with 288 GB of HBM the working sets of hundreds of frames are resident.
template <bool TABLES_IN_LDS>
__global__ void __launch_bounds__(64) k_hf_entropy_lanes(const DevPlan *plans, const HfLaneWork *work) {
	extern __shared__ __attribute__((aligned(16))) uint8_t hf_lds[];
	const HfLaneWork w = work[blockIdx.x];
	DevPlan plan = plans[w.frame];
	// pointers loaded from memory are generic to the compiler (flat_load/flat_store: slower, and they tie up both
	// the LDS and the vector-memory counters); tell it that every plan pointer is a global address
#define J40_TO_GLOBAL(p) p = (std::remove_reference<decltype((p))>::type) (__attribute__((address_space(1))) void *) (void *) (p)
	J40_TO_GLOBAL(plan.frame); J40_TO_GLOBAL(plan.codestream); J40_TO_GLOBAL(plan.pool_u8); J40_TO_GLOBAL(plan.pool_u16); J40_TO_GLOBAL(plan.pool_i32);
	J40_TO_GLOBAL(plan.pool_u64); J40_TO_GLOBAL(plan.pool_f32); J40_TO_GLOBAL(plan.clusters); J40_TO_GLOBAL(plan.coeff_specs); J40_TO_GLOBAL(plan.lf_groups);
	J40_TO_GLOBAL(plan.sections); J40_TO_GLOBAL(plan.group_blocks); J40_TO_GLOBAL(plan.group_block_start); J40_TO_GLOBAL(plan.coeffs[0]); J40_TO_GLOBAL(plan.coeffs[1]);
	J40_TO_GLOBAL(plan.coeffs[2]); J40_TO_GLOBAL(plan.nonzeros); J40_TO_GLOBAL(plan.lz_window); J40_TO_GLOBAL(plan.status);
	J40_TO_GLOBAL(plan.events); J40_TO_GLOBAL(plan.ev_range); J40_TO_GLOBAL(plan.block_events); J40_TO_GLOBAL(plan.section_end_bit);
#undef J40_TO_GLOBAL
	const DevFrame &f = *plan.frame;
	const int32_t lane = threadIdx.x;
	const bool active = lane < w.num_groups;
	const int32_t g = w.first_group + (active ? lane : 0);
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	const uint32_t n_bctx = (uint32_t) (39 * f.lfidx_size * (f.nb_qf_thr + 1));
	uint8_t *l_bctx = hf_lds;
	int16_t *l_nnz = (int16_t *) (hf_lds + align16(n_bctx));
	int8_t *l_freq = (int8_t *) (hf_lds + align16(n_bctx) + 128);
	const uint32_t off_pass = align16(n_bctx) + 192;
	{
		const uint8_t *src = plan.pool_u8 + plan.block_ctx_map_off;
		for (uint32_t i = lane; i < n_bctx; i += 64) l_bctx[i] = src[i];
		l_nnz[lane] = DEV_NNZ_CTX2[lane]; l_freq[lane] = DEV_FREQ_CTX2[lane];
	}
	HfTables t;
	t.block_ctx_map = l_bctx; t.nnz_ctx2 = l_nnz; t.freq_ctx2 = l_freq;
	t.nonzeros = plan.nonzeros + (size_t) g * (32 * 32 * 3);
	t.window = plan.lz_window ? plan.lz_window + (size_t) g * plan.lz_window_size : nullptr;
	t.blocks = plan.group_blocks + plan.group_block_start[g];
	t.nblocks = (int32_t) (plan.group_block_start[g + 1] - plan.group_block_start[g]);
	t.block_first = plan.group_block_start[g];
	t.ev_first = f.sparse_coeffs ? plan.ev_range[2 * g] : 0; t.ev_end = f.sparse_coeffs ? plan.ev_range[2 * g + 1] : 0;
	for (int32_t pass = 0; pass < f.num_passes; ++pass) {
		const DevCodeSpec &spec = plan.coeff_specs[pass];
		if (TABLES_IN_LDS) {
			__syncthreads();
			uint8_t *l_map = hf_lds + off_pass;
			DevCluster *l_clusters = (DevCluster *) (l_map + align16((uint32_t) spec.num_dist));
			uint8_t *l_tables = (uint8_t *) l_clusters + align16((uint32_t) spec.num_clusters * (uint32_t) sizeof(DevCluster));
			const uint8_t *msrc = plan.pool_u8 + spec.cluster_map_off;
			for (int32_t i = lane; i < spec.num_dist; i += 64) l_map[i] = msrc[i];
			const DevCluster *csrc = plan.clusters + spec.cluster_off;
			const uint32_t base_off = csrc[0].table_off;
			for (int32_t i = lane; i < spec.num_clusters; i += 64) { DevCluster c = csrc[i]; c.table_off -= base_off; l_clusters[i] = c; }
			if (spec.use_prefix_code) {
				const int32_t *src = plan.pool_i32 + base_off; int32_t *dst = (int32_t *) l_tables;
				for (uint32_t i = lane; i < spec.table_span; i += 64) dst[i] = src[i];
			} else {
				const uint64_t *src = plan.pool_u64 + base_off; uint64_t *dst = (uint64_t *) l_tables;
				for (uint32_t i = lane; i < spec.table_span; i += 64) dst[i] = src[i];
			}
			t.clusters = l_clusters; t.cluster_map = l_map; t.alias = (const uint64_t *) l_tables; t.prefix = (const int32_t *) l_tables;
		} else {
			t.clusters = plan.clusters + spec.cluster_off; t.cluster_map = plan.pool_u8 + spec.cluster_map_off;
			t.alias = plan.pool_u64; t.prefix = plan.pool_i32;
		}
		__syncthreads();
		if (active) {
			const DevSection &sec = plan.sections[pass * f.num_groups + g];
			plan.status[pass * f.num_groups + g] = f.sparse_coeffs ? decode_hf_section_flat<true>(plan, f, spec, t, pass, sec) : decode_hf_section_flat<false>(plan, f, spec, t, pass, sec);
		}
	}
}

// copies a POD made of 32-bit words out of global memory (address-space qualified structs have no copy constructor)
template <typename T> __device__ __forceinline__ T load_global_pod(const J40_GLOBAL T *src) {
	static_assert(sizeof(T) % 4 == 0, "word-sized PODs only");
	T out;
	const J40_GLOBAL uint32_t *p = (const J40_GLOBAL uint32_t *) src;
	uint32_t *q = (uint32_t *) &out;
#pragma unroll
	for (unsigned i = 0; i < sizeof(T) / 4; ++i) q[i] = p[i];
	return out;
}

// K1, throughput form, fast path (hf_lanes_dev.h): rANS specs without LZ77 whose packed tables fit in LDS.
// One section per wavefront LANE, blockDim.x / 64 wavefronts per workgroup sharing one copy of their frame's tables, any number of
// frames per launch. Two ways the lanes get their sections (HfLaneWork::pad):
//   0  static: lane l of the wavefront takes slot first_group + l of its frame's lane order, and is done with it;
//   1  queued: the same to begin with, and a lane that has finished a section takes the frame's next slot from a counter all the
//      frame's lanes share (`queue[frame]`, which the host starts at the number of lanes the frame was given) -- for launches with
//      more sections than the machine has lanes: the sections go out by decreasing size, so the lanes of a frame end together.
// what hands a lane its sections (hf_lanes_dev.h: decode_hf_sections_lane)
struct LaneQueue {
	const J40_GLOBAL DevSection *sections; const J40_GLOBAL DevLfGroup *lf_groups; const J40_GLOBAL uint32_t *block_start, *ev_range, *lane_order;
	J40_GLOBAL uint32_t *status, *end_bits, *counter;
	int32_t num_groups, pass, slot, g; bool scan;
	__device__ __forceinline__ bool next(LaneSection &S) {
		int32_t s = slot;
		slot = -1;
		if (s < 0) { if (!counter) return false; s = (int32_t) atomicAdd((uint32_t *) counter, 1u); }
		if (s >= num_groups) return false;
		g = lane_order ? (int32_t) lane_order[s] : s;
		const DevSection sec = load_global_pod(sections + (pass * num_groups + g));
		S.start_bit = 8u * sec.byte_off + sec.bit_off; S.end_bit = 8u * (sec.byte_off + sec.size);
		S.cell_base = (uint32_t) lf_groups[sec.ggidx].cell_base;
		S.block_first = block_start[g]; S.nblocks = (int32_t) (block_start[g + 1] - S.block_first);
		S.ev_first = scan ? ev_range[2 * g] : 0u; S.ev_end = scan ? ev_range[2 * g + 1] : 0u;
		return true;
	}
	__device__ __forceinline__ void done(uint32_t st, uint32_t end_bit) {
		status[pass * num_groups + g] = st;
		if (end_bits) end_bits[pass * num_groups + g] = end_bit;
	}
};

__global__ void __launch_bounds__(512) k_hf_lanes(const DevPlan *plans, const HfLaneWork *work, uint32_t lds_tables_bytes, uint32_t *queue) {
	extern __shared__ __attribute__((aligned(16))) uint8_t hf_lds[];
	// The launch ends with its slowest wavefront, and a wavefront runs as fast as its SIMD issues its instructions: a background
	// wavefront on the same SIMD (the LfGroup lane decoder of a later batch) took a third of them (46 -> 75 ms). Highest priority here.
	__builtin_amdgcn_s_setprio(3);
	// blockDim.x / 64 wavefronts per workgroup, all on the same frame (the host pads the work list), sharing its tables
	const int32_t tid = threadIdx.x, lane = tid & 63;
	const HfLaneWork w = work[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)];
	if (__builtin_amdgcn_readfirstlane(w.pad) & 2) __builtin_amdgcn_s_setprio(2);   // (the lighter wavefront of the two on this SIMD: async.hip)
	const J40_GLOBAL DevPlan &plan = ((const J40_GLOBAL DevPlan *) plans)[w.frame];
	const J40_GLOBAL DevFrame &df = *(const J40_GLOBAL DevFrame *) plan.frame;
	const bool active = lane < w.num_groups;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	LaneFrame f;
	f.nb_block_ctx = df.nb_block_ctx; f.num_hf_presets = df.num_hf_presets; f.preset_bits = df.preset_bits; f.check_section_end = df.check_section_end; f.single_declared_end = df.single_declared_end;
	f.order_off = df.order_off;
	const int32_t num_passes = df.num_passes, num_groups = df.num_groups, scan = df.sparse_coeffs;
	LaneGlobals G;
	G.codestream = (const J40_GLOBAL uint8_t *) plan.codestream;
	G.group_blocks = (const J40_GLOBAL uint32_t *) plan.group_blocks;
	G.coeffs = (J40_GLOBAL float *) plan.coeffs[0];
	G.events = (J40_GLOBAL CoeffEvent *) plan.events; G.block_events = (J40_GLOBAL uint32_t *) plan.block_events;
	G.pool_u16 = (const J40_GLOBAL uint16_t *) plan.pool_u16;
	G.coeff_stride = plan.coeff_stride;
	const J40_GLOBAL uint8_t *pool_u8 = (const J40_GLOBAL uint8_t *) plan.pool_u8;
	const J40_GLOBAL int32_t *pool_i32 = (const J40_GLOBAL int32_t *) plan.pool_i32;
	const J40_GLOBAL uint64_t *pool_u64 = (const J40_GLOBAL uint64_t *) plan.pool_u64;
	const J40_GLOBAL DevCodeSpec *specs = (const J40_GLOBAL DevCodeSpec *) plan.coeff_specs;
	const J40_GLOBAL DevCluster *clusters = (const J40_GLOBAL DevCluster *) plan.clusters;
	LaneQueue q;
	q.sections = (const J40_GLOBAL DevSection *) plan.sections; q.lf_groups = (const J40_GLOBAL DevLfGroup *) plan.lf_groups;
	q.block_start = (const J40_GLOBAL uint32_t *) plan.group_block_start; q.ev_range = (const J40_GLOBAL uint32_t *) plan.ev_range;
	q.lane_order = (const J40_GLOBAL uint32_t *) plan.lane_order;
	q.status = (J40_GLOBAL uint32_t *) plan.status; q.end_bits = df.sections_have_trailer ? (J40_GLOBAL uint32_t *) plan.section_end_bit : nullptr;
	// (queued lanes only in single-pass frames: the passes of a group accumulate into the same coefficients, one after the other on one lane)
	q.counter = w.pad && queue && num_passes == 1 ? (J40_GLOBAL uint32_t *) queue + w.frame : nullptr;
	q.num_groups = num_groups; q.scan = scan != 0; q.g = 0;

	J40_LDS uint8_t *lds = (J40_LDS uint8_t *) hf_lds;
	J40_LDS int16_t *l_nnz = (J40_LDS int16_t *) lds;
	J40_LDS int8_t *l_freq = (J40_LDS int8_t *) (lds + 128);
	J40_LDS uint32_t *l_dct = (J40_LDS uint32_t *) (lds + 192);
	const uint32_t off_pass = 192 + 112;
	if (tid < 64) { l_nnz[tid] = DEV_NNZ_CTX2[tid]; l_freq[tid] = DEV_FREQ_CTX2[tid]; }
	if (tid < 27) l_dct[tid] = (uint32_t) DEV_DCT_SELECT[tid][0] | ((uint32_t) DEV_DCT_SELECT[tid][1] << 8) | ((uint32_t) DEV_DCT_SELECT[tid][2] << 16);
	LaneTables t;
	t.nnz_ctx2 = l_nnz; t.freq_ctx2 = l_freq; t.dct_info = l_dct;
	// per-wave column predictor state behind the tables: [3][32][64 lanes] bytes
	J40_LDS int8_t *l_cols = (J40_LDS int8_t *) (lds + lds_tables_bytes + (uint32_t) (tid >> 6) * HF_LANE_COLS_BYTES) + lane;
	// ... and the lanes' event rings behind it: [HF_LANE_RING_SLOTS][64 lanes] words
	J40_LDS uint32_t *l_ring = (J40_LDS uint32_t *) (lds + lds_tables_bytes + (uint32_t) (tid >> 6) * HF_LANE_COLS_BYTES + HF_LANE_PRED_BYTES) + lane;
	for (int32_t pass = 0; pass < num_passes; ++pass) {
		const J40_GLOBAL DevCodeSpec &spec = specs[pass];
		__syncthreads();   // previous pass' tables are no longer in use
		const int32_t num_dist = spec.num_dist, num_clusters = spec.num_clusters, log_alpha = spec.log_alpha_size;
		J40_LDS uint8_t *l_map = lds + off_pass;
		J40_LDS uint32_t *l_cfg = (J40_LDS uint32_t *) (lds + off_pass + align16((uint32_t) num_dist));
		J40_LDS uint64_t *l_alias = (J40_LDS uint64_t *) ((J40_LDS uint8_t *) l_cfg + align16(4u * (uint32_t) num_clusters));
		{
			const J40_GLOBAL uint32_t *msrc = (const J40_GLOBAL uint32_t *) (pool_u8 + spec.cluster_map_off);   // the u8 pool keeps 4-byte alignment per table
			J40_LDS uint32_t *mdst = (J40_LDS uint32_t *) l_map;
			for (int32_t i = tid; i < (num_dist + 3) / 4; i += blockDim.x) mdst[i] = msrc[i];
			const J40_GLOBAL int32_t *csrc = pool_i32 + spec.lane_cfg_off;
			for (int32_t i = tid; i < num_clusters; i += blockDim.x) l_cfg[i] = (uint32_t) csrc[i];
			const J40_GLOBAL uint64_t *asrc = pool_u64 + clusters[spec.cluster_off].table_off;
			for (uint32_t i = tid; i < ((uint32_t) num_clusters << log_alpha); i += blockDim.x) l_alias[i] = asrc[i];
		}
		t.ctx_map = l_map; t.cluster_cfg = l_cfg; t.alias = l_alias; t.log_alpha = log_alpha; t.log_bucket = 12 - log_alpha;
		__syncthreads();
		q.pass = pass; q.slot = active ? w.first_group + lane : -1;
		if (scan) decode_hf_sections_lane<true>(f, t, G, q, l_cols, 64, pass, l_ring, 64);
		else decode_hf_sections_lane<false>(f, t, G, q, l_cols, 64, pass);
	}
}

// `queue`: one counter per frame of the batch for the queued form (HfLaneWork::pad), or null
void launch_hf_lanes(const DevPlan *plans, const HfLaneWork *work, int32_t num_work, int32_t waves_per_wg, uint32_t lds_bytes, hipStream_t stream, hipEvent_t started, hipEvent_t stopped, uint32_t *queue) {
	if (num_work <= 0) return;
	static bool configured = false;
	if (!configured) { (void) hipFuncSetAttribute((const void *) k_hf_lanes, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
	const uint32_t tables = (lds_bytes + 15u) & ~15u;
	// (started / stopped: events the device records when the kernel's first wavefront starts and its last one ends -- the kernel's own
	// duration, as rocprofv3 reports it, without the time the launch waited in its queue)
	if (started && stopped) hipExtLaunchKernelGGL(k_hf_lanes, dim3((unsigned) (num_work / waves_per_wg)), dim3(64u * (unsigned) waves_per_wg), tables + (uint32_t) waves_per_wg * HF_LANE_COLS_BYTES, stream, started, stopped, 0, plans, work, tables, queue);
	else hipLaunchKernelGGL(k_hf_lanes, dim3((unsigned) (num_work / waves_per_wg)), dim3(64u * (unsigned) waves_per_wg), tables + (uint32_t) waves_per_wg * HF_LANE_COLS_BYTES, stream, plans, work, tables, queue);
}

// LDS bytes k_hf_entropy_lanes needs for one frame
uint32_t hf_lanes_lds_bytes(const HfLaunchInfo &info) {
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	uint32_t n = align16(info.block_ctx_size) + 192;
	if (info.tables_fit_lds) n += align16(info.max_num_dist) + align16(info.max_clusters * (uint32_t) sizeof(DevCluster)) + align16(info.max_table_bytes);
	return n;
}

void launch_hf_entropy_lanes(const DevPlan *plans, const HfLaneWork *work, int32_t num_work, bool tables_in_lds, uint32_t lds_bytes, hipStream_t stream) {
	if (num_work <= 0) return;
	if (tables_in_lds) {
		static bool configured = false;
		if (!configured) { (void) hipFuncSetAttribute((const void *) k_hf_entropy_lanes<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
		hipLaunchKernelGGL(k_hf_entropy_lanes<true>, dim3((unsigned) num_work), dim3(64), lds_bytes, stream, plans, work);
	} else {
		hipLaunchKernelGGL(k_hf_entropy_lanes<false>, dim3((unsigned) num_work), dim3(64), lds_bytes, stream, plans, work);
	}
}

// ------------------------------------------------------------------------------------------------
// K2 common pieces

// Instrumented build (-DJ40_K2_PHASES, tools/build_variant.sh; never the product's): lane 0 of every workgroup of k_vardct_dct adds up
// the shader clock (s_memtime) it spends in each phase of a tile -- as wave 0 sees it: a phase ends behind its barrier -- and adds the
// sums to g_k2_phase[shape][phase] when it is done; j40hip_debug_k2_phases copies the table out. Phases: 0 prologue (bind, geometry,
// event prefix), 1 zeroing + barrier, 2 event scatter + LLF + barrier, 3 pass 1 + barrier, 4 pass 2 + barrier, 5 colour + stores,
// 6 the barrier behind them, 7 tiles counted.
#ifdef J40_K2_PHASES
__device__ unsigned long long g_k2_phase[64][8];
#define K2_PHASES_BEGIN unsigned long long k2_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long k2_last = __builtin_amdgcn_s_memtime()
#define K2_PHASE(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); k2_acc[i] += now_ - k2_last; k2_last = now_; } while (0)
#define K2_PHASES_END(slot) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_k2_phase[slot][i_], k2_acc[i_]); } while (0)
extern "C" __attribute__((visibility("default"))) void j40hip_debug_k2_phases(unsigned long long *out, int reset) {
	(void) hipDeviceSynchronize();
	(void) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k2_phase), sizeof(unsigned long long) * 64 * 8, 0, hipMemcpyDeviceToHost);
	if (reset) { static unsigned long long zero[64 * 8]; (void) hipMemcpyToSymbol(HIP_SYMBOL(g_k2_phase), zero, sizeof zero, 0, hipMemcpyHostToDevice); }
}
#else
#define K2_PHASES_BEGIN do { } while (0)
#define K2_PHASE(i) do { } while (0)
#define K2_PHASES_END(slot) do { } while (0)
#endif

// batch-wide launches (BATCH) are persistent: a launch covers one class of transforms over every frame of the batch, its work
// cut into tiles of `per_wg` varblocks; tile_prefix[f] = tiles of the frames before frame f (built on the device by k_k2_tiles
// from the frames' class_start, which the device-side plan build writes: the host never learns the counts). Workgroup b takes
// a contiguous run of tiles (one search for its first tile's frame, then it walks along); k2_bind replaces the kernel arguments with
// the next tile's frame's list, count and output when the run enters another frame; `first` = the tile's first varblock. Returns
// false when the run is done.
//
// What a frame's tiles share (its plan's pointers, colour constants, lists) is the same in every lane; said so to the compiler
// (uni(): v_readfirstlane), it lives in scalar registers across the tiles of a frame instead of fifty vector registers per lane.
__device__ __forceinline__ void uniform_plan(DevPlan &p) {   // (the fields the pixel kernels read)
	p.frame = uni(p.frame); p.pool_u16 = uni(p.pool_u16); p.pool_f32 = uni(p.pool_f32); p.events = uni(p.events); p.block_events = uni(p.block_events);
	for (int c = 0; c < 3; ++c) { p.llf[c] = uni(p.llf[c]); p.coeffs[c] = uni(p.coeffs[c]); }
	p.coeff_stride = uni(p.coeff_stride);
}
__device__ __forceinline__ void uniform_colour(ColourConsts &c) {
	for (int i = 0; i < 3; ++i) { c.cbrt_opsin_bias[i] = uni(c.cbrt_opsin_bias[i]); c.opsin_bias[i] = uni(c.opsin_bias[i]); }
	for (int i = 0; i < 9; ++i) c.m[i] = uni(c.m[i]);
	c.itscale = uni(c.itscale); c.bpp = uni(c.bpp);
}

// (K2Iter, k2_run_begin, k2_run_bind: k2_iter_dev.h -- also compiled for the CPU, where tests/hostsim walks every workgroup's run)
template <bool BATCH>
__device__ __forceinline__ K2Iter k2_begin(const int32_t *tile_prefix, int32_t nframes) {
	if (!BATCH) return K2Iter{0, 1, 0, 0, 0};
	return k2_run_begin(tile_prefix, nframes, (int32_t) blockIdx.x, (int32_t) gridDim.x);
}
template <bool BATCH>
__device__ __forceinline__ bool k2_bind(K2Iter &it, const K2Frame *batch, const int32_t *tile_prefix, int32_t class_a, int32_t class_b, int32_t per_wg,
		const DevVarblock *&list, int32_t &count, uint8_t *&rgba, size_t &stride, int32_t &frame, int32_t &first, bool &entered) {
	if (!BATCH) { frame = 0; first = (int32_t) blockIdx.x * per_wg; entered = it.tile == 0; return it.tile++ == 0; }
	return k2_run_bind(it, batch, tile_prefix, class_a, class_b, per_wg, list, count, rgba, stride, frame, first, entered);
}

// clears `n` floats of LDS tiles (n a multiple of four, the tiles 16-byte aligned: every tile size here is) with 16-byte stores: a
// quarter of the LDS instructions of a float at a time (J40_K2_ZERO_SCALAR: the older loop, for comparison)
__device__ __forceinline__ void zero_tiles(float *t, int32_t n, int32_t tid, int32_t nthreads) {
#ifdef J40_K2_ZERO_SCALAR
	for (int32_t w = tid; w < n; w += nthreads) t[w] = 0.0f;
#else
	typedef float f4 __attribute__((ext_vector_type(4)));
	const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
	for (int32_t w = 4 * tid; w < n; w += 4 * nthreads) *(f4 *) (t + w) = z;
#endif
}

// exclusive prefix sums of the per-block event counts held by lanes 0..NB-1 of the first wavefront (`mine`, 0 for lanes
// past the last block) -> prefix[0..NB]; visible to the workgroup after its next barrier
template <int NB>
__device__ __forceinline__ void stage_event_prefix(uint32_t mine, uint32_t *prefix, int32_t tid) {
	if (tid >= 64) return;
	uint32_t incl = mine;
#pragma unroll
	for (int d = 1; d < NB; d <<= 1) { const uint32_t up = __shfl_up(incl, d); if (tid >= d) incl += up; }
	if (tid < NB) prefix[tid + 1] = incl;
	if (tid == 0) prefix[0] = 0;
}

// ------------------------------------------------------------------------------------------------
// K2: DCT family up to 64x64. LDS tile per (varblock, channel): rows x (columns + 1) floats, element
// (r, c) = vertical frequency r, horizontal frequency c. Pass 1: one lane per row r transforms along
// c in registers; pass 2: one lane per column x transforms along r. Both walk LDS conflict-free
// thanks to the odd pitch. Arithmetic = j40__inverse_dct2d (j40.h:5972): IDCT over the columns
// dimension first, then over rows.

// The run's next tile, when it lies in the same frame (k2_run_bind has stepped `it.tile` to it), starts NB records further on: lanes
// 0 .. NB - 1 fetch the ordinals (`blk`) of its blocks now, one word each, and have them a whole tile's work later -- the next
// prologue then asks for a block's record and for its entry of block_events side by side (J40_K2_NO_PREFETCH: as before, the entry
// after the record)
// ---- J40_K2_AHEAD (the default): the records of the run's next TWO tiles and the block_events entries of its next tile are on their
// way while a tile is worked on, and they travel without registers: global_load_lds_dword has the memory system write a lane's dword
// to LDS at M0 + 4 * lane. A tile's prologue was a round trip to memory for its records (with the entry of block_events beside it
// once the ordinals were prefetched) in front of the round trip for its events -- a quarter of the 8x8 kernel's time per tile (the
// instrumented build, call M) with every wavefront slot of the machine taken, so a slot's time is what the stage costs. Now:
//   tile k, first thing, wavefront 0: asks for the records of tile k + 2 (ring of three slots) and, with the ordinals of tile k + 1's
//   records (asked for during tile k - 1, in LDS since), for the entries of tile k + 1 (ring of two);
//   wavefront 0 waits for its counter (vmcnt(0)) just before the barrier behind the scatter phase -- its own event loads have come
//   back by then, so the wait is short -- and that barrier publishes what has arrived to the other wavefronts;
//   tile k itself finds its records and entries in LDS.
// Where a run starts or enters another frame nothing is on its way: wavefront 0 fetches the tile's own records and entries the same
// way and waits for them (the two round trips the prologue always had). The compiler does not know the instruction (inline assembly:
// told about it, it orders every later LDS read of the same array behind the copy, i.e. behind the round trip): the waits are the
// two above, by hand. M0 is not used by anything else in these kernels.
#ifndef J40_K2_AHEAD
// (global_load_lds_dword is a gfx9 instruction: the library is written for gfx950; built for anything else -- the Makefile's ARCH can be
// overridden -- the prologue falls back to K2_PREFETCH_BLK)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx940__) && !defined(__gfx90a__)
#define J40_K2_AHEAD 0
#else
#define J40_K2_AHEAD 1
#endif
#endif
static_assert(sizeof(DevVarblock) == 40, "K2Ahead copies records as ten dwords");
__device__ __forceinline__ void k2_copy_dword_to_lds(const uint32_t *src_of_lane, uint32_t lds_byte_address) {
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src_of_lane), "s"(lds_byte_address) : "memory");   // (m0 cannot be named as clobbered: it is reserved to the compiler, which sets it itself only for movrel / GWS / LDS-direct -- none of which these kernels use)
}
__device__ __forceinline__ void k2_copies_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int NB> struct K2Ahead {
	// (wavefront 0 alone issues the copies and waits for them, and lanes 0 .. nb - 1 read the records back without a barrier: they have to be its lanes)
	static_assert(NB <= 64, "K2Ahead: a tile's blocks must fit wavefront 0");
	enum { REC_DW = 10, REC_SLOT = NB * REC_DW, BE_SLOT = NB * 4 };
	uint32_t *rec, *be;   // LDS: [3][REC_SLOT], [2][BE_SLOT]
	__device__ __forceinline__ static uint32_t lds_address(const uint32_t *p) { return (uint32_t) __builtin_amdgcn_readfirstlane((int32_t) (uint32_t) (uintptr_t) p); }   // (the low half of a flat LDS address is the LDS offset)
	__device__ __forceinline__ const uint32_t *records(int32_t k) const { return rec + (k % 3) * REC_SLOT; }
	__device__ __forceinline__ const uint32_t *entries(int32_t k) const { return be + (k & 1) * BE_SLOT; }
	// wavefront 0, lane = 0 .. 63: tile k of `list` (its blocks k * NB ...)
	__device__ __forceinline__ void ask_records(const DevVarblock *list, int32_t count, int32_t k, int32_t lane) const {
		const int32_t first = k * NB, n = min(NB, count - first) * REC_DW;
		const uint32_t *src = (const uint32_t *) (list + first);
		const uint32_t base = lds_address(records(k));
#pragma unroll
		for (int32_t j = 0; j * 64 < REC_SLOT; ++j) { const int32_t d = j * 64 + lane; if (d < n) k2_copy_dword_to_lds(src + d, base + (uint32_t) j * 256u); }
	}
	// ... the entries of block_events of tile k's `nb` blocks; the tile's records are in LDS
	__device__ __forceinline__ void ask_entries(const uint32_t *block_events, int32_t nb, int32_t k, int32_t lane) const {
		const uint32_t *r = records(k);
		const uint32_t base = lds_address(entries(k));
#pragma unroll
		for (int32_t j = 0; j * 64 < BE_SLOT; ++j) {
			const int32_t l = j * 64 + lane;
			if (l < nb * 4) { const uint32_t blk = r[(l >> 2) * REC_DW + 8]; k2_copy_dword_to_lds(block_events + 4 * (size_t) blk + (uint32_t) (l & 3), base + (uint32_t) j * 256u); }
		}
	}
};
// the tile's prologue for both kernel families: `k` = the tile's number in its frame's list; true: records and entries are in LDS
#define K2_AHEAD_PROLOGUE(NB_) \
	const int32_t ahead_k = first / (NB_); \
	if (BATCH && J40_K2_AHEAD) { \
		const bool next_ok = it.tile < it.tile_end && it.tile < it.frame_end, next2_ok = it.tile + 1 < it.tile_end && it.tile + 1 < it.frame_end; \
		if (tid < 64) { \
			if (entered) { \
				ahead.ask_records(list, count, ahead_k, tid); \
				if (next_ok) ahead.ask_records(list, count, ahead_k + 1, tid); \
				k2_copies_wait(); \
				if (sparse) { ahead.ask_entries(plan.block_events, nb, ahead_k, tid); k2_copies_wait(); } \
			} \
			if (next2_ok) ahead.ask_records(list, count, ahead_k + 2, tid); \
			if (next_ok && sparse) ahead.ask_entries(plan.block_events, min((NB_), count - first - (NB_)), ahead_k + 1, tid); \
		} \
	}

#ifdef J40_K2_NO_PREFETCH
#define K2_PREFETCH_BLK(NB_) do { next_blk_valid = false; } while (0)
#else
#define K2_PREFETCH_BLK(NB_) do { \
		next_blk_valid = BATCH && it.tile < it.tile_end && it.tile < it.frame_end; \
		if (next_blk_valid && first + (NB_) + tid < count && tid < (NB_)) next_blk = list[first + (NB_) + tid].blk; \
	} while (0)
#endif

// J40_K2_WAVES_PER_EU: the kernels of the small shapes (up to 16 x 8) are asked to fit the registers that eight wavefronts per SIMD
// leave -- 53-58 instead of 72-77, no spills --, which pays once the prologue's two loads travel together: pixel stage of 256 8K frames
// alone 58.8-59.2 ms without the prefetch, 57.6-58.7 with it, 56.6-56.8 with both (profiles/r05_ab_k2_prefetch_next_tile_blk_call_h.jsonl)
#ifndef J40_K2_WAVES_PER_EU
#define J40_K2_WAVES_PER_EU 8
#endif
// XYB = true (single-frame launches only; the restoration filters' input, restore_kernels.hip): the samples leave as they come out of
// the inverse transforms -- three float planes of the frame, X, Y, B, `stride_bytes` per row (4 bytes a sample, like RGBA), one
// behind the other from `rgba` -- instead of going through the colour conversion. The fused default is not touched by it.
__device__ __forceinline__ void store_xyb(uint8_t *base, size_t off, size_t plane_bytes, float sx, float sy, float sb) {
	*(float *) (base + off) = sx; *(float *) (base + plane_bytes + off) = sy; *(float *) (base + 2 * plane_bytes + off) = sb;
}

template <int LOGR, int LOGC, int NB, bool BATCH, bool XYB = false>
__global__ void __launch_bounds__(256, (LOGR + LOGC <= 7 ? J40_K2_WAVES_PER_EU : 1)) k_vardct_dct(DevPlan plan_arg, const DevVarblock *list, int32_t count, int32_t param_idx, int32_t order_idx, uint8_t *rgba, size_t stride_bytes,
		const K2Frame *batch, const int32_t *tile_prefix, int32_t nframes, int32_t class_a, int32_t class_b) {
	constexpr int R = 1 << LOGR, C = 1 << LOGC, P = C + 1, TILE = R * P;
	constexpr int LONG = R > C ? R : C;                  // columns of the canonical (short side = rows) layout
	constexpr int VH8 = (R < C ? R : C) / 8, VW8 = LONG / 8;
	extern __shared__ __attribute__((aligned(16))) float lds[];   // [NB][3][TILE]
	const int32_t tid = threadIdx.x, nthreads = blockDim.x;
	__shared__ VbGeom geom[NB];
	__shared__ uint32_t g_be[NB][4];   // each block's entry of DevPlan::block_events
	__shared__ size_t g_out[NB];       // byte offset of each block's top-left pixel in the output
	__shared__ uint32_t ev_prefix[NB + 1];   // events of the blocks before each block (tiles_scatter_events)
	J40_STAGE_SRGB_THRESHOLDS(f);
	constexpr int N = R * C;
	constexpr int PAR = N >= 256 ? 1 : 256 / N;   // blocks a pass of the 256 lanes covers
	constexpr int PER = N >= 256 ? N / 256 : 1;   // pixel positions per lane
	K2_PHASES_BEGIN;
	// what a frame's tiles share, fetched when the workgroup's run of tiles enters the frame (wave-uniform: scalar registers)
	DevPlan plan = plan_arg;
	ColourConsts cc;
	const float *dq = nullptr, *dq_scan = nullptr; const uint16_t *order = nullptr;
	float qbias0 = 0, qbias1 = 0, qbias2 = 0, qbias_num = 0, kx_lf = 0, kb_lf = 0, x_qm_mul = 0, b_qm_mul = 0;
	bool sparse = false;
	int32_t next_blk = 0; bool next_blk_valid = false;   // lane b: the ordinal of block b of the run's NEXT tile, fetched while this one is worked on (J40_K2_AHEAD=0)
	__shared__ uint32_t ahead_rec[3 * K2Ahead<NB>::REC_SLOT], ahead_be[2 * K2Ahead<NB>::BE_SLOT];
	const K2Ahead<NB> ahead = {ahead_rec, ahead_be};
	for (K2Iter it = k2_begin<BATCH>(tile_prefix, nframes); ; ) {
		int32_t frame, first; bool entered;
		if (!k2_bind<BATCH>(it, batch, tile_prefix, class_a, class_b, NB, list, count, rgba, stride_bytes, frame, first, entered)) break;
		if (entered) {
			if (BATCH) { plan = batch[frame].plan; uniform_plan(plan); }
			const DevFrame &f = *plan.frame;
			cc = load_colour_consts(f); uniform_colour(cc);
			dq = uni(plan.pool_f32 + f.dq_off[param_idx]);
			order = uni(plan.pool_u16 + f.order_off[order_idx * 3]);   // pass 0; the three channels' orders are consecutive
			dq_scan = uni(plan.pool_f32 + f.dq_scan_off[param_idx]);
			qbias0 = uni(f.quant_bias[0]); qbias1 = uni(f.quant_bias[1]); qbias2 = uni(f.quant_bias[2]); qbias_num = uni(f.quant_bias_num); kx_lf = uni(f.kx_lf); kb_lf = uni(f.kb_lf);
			x_qm_mul = uni(f.x_qm_mul); b_qm_mul = uni(f.b_qm_mul); sparse = uni((int32_t) f.sparse_coeffs) != 0;
		}
		const int32_t nb = min(NB, count - first);
		const int32_t dq_size = R * C;
		// (asking for the tile's records first and zeroing the tiles while they are on their way was measured: nineteen registers more
		// and 67.2 against 64.4 ms for the stage)
		K2_AHEAD_PROLOGUE(NB)
		if (sparse) zero_tiles(lds, nb * 3 * TILE, tid, nthreads);
		uint32_t nevents = 0;
		if (tid < nb) {
			DevVarblock vb;
			uint32_t be0 = 0, be1 = 0, be2 = 0, be3 = 0;
			if (BATCH && J40_K2_AHEAD) {   // record and entry are in LDS (K2Ahead)
				vb = *(const DevVarblock *) (ahead.records(ahead_k) + tid * 10);
				if (sparse) { const uint32_t *be = ahead.entries(ahead_k) + tid * 4; be0 = be[0]; be1 = be[1]; be2 = be[2]; be3 = be[3]; }
			} else {
				vb = list[first + tid];
				// (the block's entry of block_events is asked for together with its record when the tile before this one fetched its
				// ordinal ahead -- K2_PREFETCH_BLK below --: one round trip to memory in front of the tile instead of two in a row)
				const int32_t blk = next_blk_valid ? next_blk : vb.blk;
				if (sparse) { const uint32_t *be = plan.block_events + 4 * (size_t) blk; be0 = be[0]; be1 = be[1]; be2 = be[2]; be3 = be[3]; }
			}
			VbGeom g;
			g.coeff_base = vb.coeff_base; g.llf_base = vb.llf_base;
			g.mult[1] = vb.mult1; g.mult[0] = vb.mult1 * x_qm_mul; g.mult[2] = vb.mult1 * b_qm_mul;   // (varblock_geometry with the frame's factors at hand; j40.h:7078-7080)
			g.kx_hf = vb.kx_hf; g.kb_hf = vb.kb_hf; g.px = vb.px; g.py = vb.py; g.effw = vb.effw; g.effh = vb.effh;
			geom[tid] = g; g_out[tid] = (size_t) g.py * stride_bytes + (size_t) g.px * 4;
			if (sparse) { g_be[tid][0] = be0; g_be[tid][1] = be1; g_be[tid][2] = be2; g_be[tid][3] = be3; nevents = be1 + be2 + be3; }
		}
		if (!(BATCH && J40_K2_AHEAD)) K2_PREFETCH_BLK(NB);
		stage_event_prefix<NB>(nevents, ev_prefix, tid);
		K2_PHASE(0);
		// ---- load: dequantise + chroma-from-luma + LLF into the LDS tiles ----
		if (sparse) {
			// single-pass frames: the tiles are zeroed (above), then the blocks' coefficient events are scattered into them (one lane per
			// event over the whole workgroup: the work is proportional to the non-zeros; chroma-from-luma rides along) and the LLF
			// corners written (vardct_dev.h). No event lands in an LLF corner, so the two need no barrier between them.
			__syncthreads();
			K2_PHASE(1);
			const TileMap map = {R, C, P, 0};
			const float qbias[3] = {qbias0, qbias1, qbias2};
			tiles_scatter_events<NB>(plan, geom, g_be, ev_prefix, order, dq_scan, nullptr, N, map, lds, 3 * TILE, TILE, qbias, qbias_num, tid, nthreads);
			tiles_fill_llf(plan, geom, nb, LONG, VH8, VW8, map, lds, 3 * TILE, TILE, kx_lf, kb_lf, tid, nthreads);
		} else {
			// multi-pass frames: dense planes in canonical order, coalesced over the canonical index
			__syncthreads();
			for (int32_t w = tid; w < nb * R * C; w += nthreads) {
				const int32_t b = w / (R * C), i = w - b * (R * C);
				const VbGeom &g = geom[b];
				float v[3];
				load_coeff3(plan, g, dq, dq_size, i, LONG, VH8, VW8, v);
				const int32_t r = C > R ? i / C : i % R, c = C > R ? i % C : i / R;
				float *t = lds + (size_t) b * 3 * TILE + r * P + c;
				t[0] = v[0]; t[TILE] = v[1]; t[2 * TILE] = v[2];
			}
		}
		if (BATCH && J40_K2_AHEAD && tid < 64) k2_copies_wait();   // (what wavefront 0 asked for at the tile's start has arrived: K2Ahead)
		__syncthreads();
		K2_PHASE(2);
		// ---- pass 1: IDCT of length C along c, one lane per (block, channel, r) ----
		for (int32_t w = tid; w < nb * 3 * R; w += nthreads) {
			float *row = lds + (size_t) (w / R) * TILE + (w % R) * P;
			float x[C];
#pragma unroll
			for (int k = 0; k < C; ++k) x[k] = row[k];
			Idct1D<C>::run(x, c_half_secants);
#pragma unroll
			for (int k = 0; k < C; ++k) row[k] = x[k];
		}
		__syncthreads();
		K2_PHASE(3);
		// ---- pass 2: IDCT of length R along r, one lane per (block, channel, x) ----
		for (int32_t w = tid; w < nb * 3 * C; w += nthreads) {
			float *col = lds + (size_t) (w / C) * TILE + (w % C);
			float x[R];
#pragma unroll
			for (int k = 0; k < R; ++k) x[k] = col[k * P];
			Idct1D<R>::run(x, c_half_secants);
#pragma unroll
			for (int k = 0; k < R; ++k) col[k * P] = x[k];
		}
		__syncthreads();
		K2_PHASE(4);
		// ---- colour + pack: a lane owns pixel position (y, x) for every block of the workgroup ----
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int32_t p = N >= 256 ? tid + 256 * k : tid % N;
			const int32_t y = p / C, x = p % C;
			const uint32_t in_block = (uint32_t) y * (uint32_t) stride_bytes + (uint32_t) x * 4u;   // a block spans < 4 GB of output
			for (int32_t b = N >= 256 ? 0 : tid / N; b < nb; b += PAR) {
				const VbGeom &g = geom[b];
				if (y >= g.effh || x >= g.effw) continue;
				const float *t = lds + (size_t) b * 3 * TILE + y * P + x;
				if constexpr (XYB) { store_xyb(rgba, g_out[b] + in_block, stride_bytes * (size_t) plan.frame->height, t[0], t[TILE], t[2 * TILE]); continue; }
				const uint32_t px = xyb_to_rgba8(t[0], t[TILE], t[2 * TILE], cc, srgb_thr);
				__builtin_nontemporal_store(px, (uint32_t *) (rgba + g_out[b] + in_block));   // written once, never read here: keep it out of the L2's way (-2 %)
			}
		}
		K2_PHASE(5);
		if (!BATCH) break;
		__syncthreads();   // the next tile reuses geom / the LDS tiles
		K2_PHASE(6);
#ifdef J40_K2_PHASES
		++k2_acc[7];
#endif
	}
	K2_PHASES_END(LOGR * 8 + LOGC);
}

// ------------------------------------------------------------------------------------------------
// K2s: the 8x8 special transforms (special8_dev.h), cooperative: eight lanes per (block, channel) tile, two phases, in place.
// The 256 lanes of a workgroup are 32 tiles x 8 lanes; a workgroup's NB blocks x 3 channels = 96 tiles take three rounds.
// Tiles are 72 floats apart with rows of 9 (conflict-free along rows and down columns: special8_dev.h).

// J40_K2_SPECIAL_NB blocks per tile, J40_K2_SPECIAL_THREADS lanes per workgroup, J40_K2_SPECIAL_WAVES wavefronts per SIMD asked of the
// register allocator. A tile's time is mostly waiting (events and LLF: 42-53 %, the stores 27 %: the instrumented build's clocks,
// call M), so what a compute unit gets done is how many workgroups it holds: 32 blocks x 3 x 72 floats = 32.8 KB of LDS and 145
// registers held it to three workgroups of four wavefronts.
#ifndef J40_K2_SPECIAL_NB
#define J40_K2_SPECIAL_NB 16
#endif
#ifndef J40_K2_SPECIAL_THREADS
#define J40_K2_SPECIAL_THREADS 256
#endif
#ifndef J40_K2_SPECIAL_WAVES
#define J40_K2_SPECIAL_WAVES 8
#endif
#ifndef J40_K2_SPECIAL_WAVES_AFV
#define J40_K2_SPECIAL_WAVES_AFV 5
#endif
// SET: which of the nine transforms the instantiation carries -- 1: DctSelect 1-3 (Hornuss, DCT2x2, DCT4x4), 2: 12-13 (the two halves
// forms), 3: 14-17 (AFV) -- a launch each: with all nine inlined into one kernel the AFV paths' 145 registers set the occupancy of all.
template <int SET> __device__ __forceinline__ void special8_set_phase0(int sel, int lane, const float *src, float *dst, const float *hs, const float *afv_basis) {
	if (SET == 1) { if (sel == 1) hornuss_phase0(lane, src, dst); else if (sel == 2) pyramid_phase0(lane, src, dst, true); else quadrants_phase0(lane, src, dst, hs); }
	else if (SET == 2) { if (sel == 12) wide_halves_phase0(lane, src, dst, hs); else tall_halves_phase0(lane, src, dst, hs); }
	else afv_phase0(lane, src, dst, hs, afv_basis);
}
template <int SET> __device__ __forceinline__ void special8_set_phase1(int sel, int lane, const float *mid, float *dst, const float *hs) {
	if (SET == 1) { if (sel == 1) copy_row_phase1(lane, mid, dst, true); else if (sel == 2) pyramid_phase1(lane, mid, dst); else quadrants_phase1(lane, mid, dst, hs); }
	else if (SET == 2) { if (sel == 12) wide_halves_phase1(lane, mid, dst, hs); else tall_halves_phase1(lane, mid, dst, hs); }
	else afv_phase1(lane, mid, dst, hs, (sel - 14) & 1, (sel - 14) >> 1);
}
template <int NB, bool BATCH, int SET, bool XYB = false>
__device__ __forceinline__ void vardct_special_body(DevPlan plan_arg, const DevVarblock *list, int32_t count, uint8_t *rgba, size_t stride_bytes, const K2Frame *batch, const int32_t *tile_prefix, int32_t nframes,
		int32_t class_a, int32_t class_b) {
	constexpr int P = SP8_TILE;
	__shared__ __attribute__((aligned(16))) float tiles[NB * 3 * P];   // coefficients in, samples out: both phases work in place (special8_dev.h)
	const int32_t tid = threadIdx.x, nthreads = blockDim.x;
	__shared__ VbGeom geom[NB];
	J40_STAGE_SRGB_THRESHOLDS(f);
	__shared__ int32_t g_param[NB], g_sel[NB];
	__shared__ uint32_t g_be[NB][4], g_dq[NB], ev_prefix[NB + 1];
	// what a frame's tiles share, fetched when the workgroup's run of tiles enters the frame (k_vardct_dct)
	DevPlan plan = plan_arg;
	ColourConsts cc;
	const uint16_t *order = nullptr;
	float qbias[3] = {0, 0, 0}, qbias_num = 0, kx_lf = 0, kb_lf = 0, x_qm_mul = 0, b_qm_mul = 0;
	uint32_t dq_scan_off[5] = {0, 0, 0, 0, 0};   // of the parameter sets 1, 2, 3, 9, 10
	bool sparse = false;
	int32_t next_blk = 0; bool next_blk_valid = false;   // (k_vardct_dct: the next tile's block ordinals, fetched ahead)
	__shared__ uint32_t ahead_rec[3 * K2Ahead<NB>::REC_SLOT], ahead_be[2 * K2Ahead<NB>::BE_SLOT];
	const K2Ahead<NB> ahead = {ahead_rec, ahead_be};
	K2_PHASES_BEGIN;
	for (K2Iter it = k2_begin<BATCH>(tile_prefix, nframes); ; ) {
		int32_t frame, first; bool entered;
		if (!k2_bind<BATCH>(it, batch, tile_prefix, class_a, class_b, NB, list, count, rgba, stride_bytes, frame, first, entered)) break;
		if (entered) {
			if (BATCH) { plan = batch[frame].plan; uniform_plan(plan); }
			const DevFrame &fe = *plan.frame;
			cc = load_colour_consts(fe); uniform_colour(cc);
			order = uni(plan.pool_u16 + fe.order_off[1 * 3]);   // all 8x8 specials share order 1
			qbias[0] = uni(fe.quant_bias[0]); qbias[1] = uni(fe.quant_bias[1]); qbias[2] = uni(fe.quant_bias[2]); qbias_num = uni(fe.quant_bias_num); kx_lf = uni(fe.kx_lf); kb_lf = uni(fe.kb_lf);
			x_qm_mul = uni(fe.x_qm_mul); b_qm_mul = uni(fe.b_qm_mul); sparse = uni((int32_t) fe.sparse_coeffs) != 0;
			dq_scan_off[0] = uni((uint32_t) fe.dq_scan_off[1]); dq_scan_off[1] = uni((uint32_t) fe.dq_scan_off[2]); dq_scan_off[2] = uni((uint32_t) fe.dq_scan_off[3]);
			dq_scan_off[3] = uni((uint32_t) fe.dq_scan_off[9]); dq_scan_off[4] = uni((uint32_t) fe.dq_scan_off[10]);
		}
		const DevFrame &f = *plan.frame;
		const int32_t nb = min(NB, count - first);
		K2_PHASE(0);
		K2_AHEAD_PROLOGUE(NB)
		if (sparse) zero_tiles(tiles, nb * 3 * P, tid, nthreads);
		K2_PHASE(1);
		uint32_t nevents = 0;
		if (tid < nb) {
			DevVarblock vb;
			uint32_t be0 = 0, be1 = 0, be2 = 0, be3 = 0;
			if (BATCH && J40_K2_AHEAD) {   // (k_vardct_dct)
				vb = *(const DevVarblock *) (ahead.records(ahead_k) + tid * 10);
				if (sparse) { const uint32_t *be = ahead.entries(ahead_k) + tid * 4; be0 = be[0]; be1 = be[1]; be2 = be[2]; be3 = be[3]; }
			} else {
				vb = list[first + tid];
				const int32_t blk = next_blk_valid ? next_blk : vb.blk;
				if (sparse) { const uint32_t *be = plan.block_events + 4 * (size_t) blk; be0 = be[0]; be1 = be[1]; be2 = be[2]; be3 = be[3]; }
			}
			VbGeom g;
			g.coeff_base = vb.coeff_base; g.llf_base = vb.llf_base;
			g.mult[1] = vb.mult1; g.mult[0] = vb.mult1 * x_qm_mul; g.mult[2] = vb.mult1 * b_qm_mul;   // (varblock_geometry; j40.h:7078-7080)
			g.kx_hf = vb.kx_hf; g.kb_hf = vb.kb_hf; g.px = vb.px; g.py = vb.py; g.effw = vb.effw; g.effh = vb.effh;
			geom[tid] = g;
			if (sparse) { g_be[tid][0] = be0; g_be[tid][1] = be1; g_be[tid][2] = be2; g_be[tid][3] = be3; nevents = be1 + be2 + be3; }
			g_sel[tid] = vb.dctsel;
			g_param[tid] = vb.dctsel == 1 ? 1 : vb.dctsel == 2 ? 2 : vb.dctsel == 3 ? 3 : vb.dctsel <= 13 ? 9 : 10;
			g_dq[tid] = vb.dctsel == 1 ? dq_scan_off[0] : vb.dctsel == 2 ? dq_scan_off[1] : vb.dctsel == 3 ? dq_scan_off[2] : vb.dctsel <= 13 ? dq_scan_off[3] : dq_scan_off[4];
		}
		if (!(BATCH && J40_K2_AHEAD)) K2_PREFETCH_BLK(NB);
		stage_event_prefix<NB>(nevents, ev_prefix, tid);
		if (sparse) {
			__syncthreads();
			const TileMap map = {8, 8, SP8_PITCH, 1};
			tiles_scatter_events<NB>(plan, geom, g_be, ev_prefix, order, plan.pool_f32, g_dq, 64, map, tiles, 3 * P, P, qbias, qbias_num, tid, nthreads);
			tiles_fill_llf(plan, geom, nb, 8, 1, 1, map, tiles, 3 * P, P, kx_lf, kb_lf, tid, nthreads);
		} else {
			__syncthreads();
			for (int32_t w = tid; w < nb * 64; w += nthreads) {
				const int32_t b = w >> 6, i = w & 63;
				float v[3];
				load_coeff3(plan, geom[b], plan.pool_f32 + f.dq_off[g_param[b]], 64, i, 8, 1, 1, v);
				float *t = tiles + (size_t) b * 3 * P + SP8(i);
				t[0] = v[0]; t[P] = v[1]; t[2 * P] = v[2];
			}
		}
		if (BATCH && J40_K2_AHEAD && tid < 64) k2_copies_wait();   // (K2Ahead)
		__syncthreads();
		K2_PHASE(2);
		// eight lanes per tile, the eight tiles of a wavefront side by side; a tile's lanes sit in one wavefront, so the two phases
		// need no workgroup barrier between them, only their order (SP8_LOADS_DONE)
		const int32_t lane8 = tid & 7;
		for (int32_t tile = tid >> 3; tile < nb * 3; tile += nthreads >> 3) {
			float *t = tiles + tile * P;
			const int32_t sel = g_sel[tile / 3];
			special8_set_phase0<SET>(sel, lane8, (const float *) t, t, c_half_secants, c_afv_basis);
			SP8_LOADS_DONE();
			special8_set_phase1<SET>(sel, lane8, (const float *) t, t, c_half_secants);
		}
		__syncthreads();
		K2_PHASE(3);
		for (int32_t w = tid; w < nb * 64; w += nthreads) {
			const int32_t b = w >> 6, i = w & 63, y = i >> 3, x = i & 7;
			const VbGeom &g = geom[b];
			if (y >= g.effh || x >= g.effw) continue;
			const float *t = tiles + (size_t) b * 3 * P + SP8(i);
			if constexpr (XYB) { store_xyb(rgba, (size_t) (g.py + y) * stride_bytes + (size_t) (g.px + x) * 4, stride_bytes * (size_t) f.height, t[0], t[P], t[2 * P]); continue; }
			const uint32_t px = xyb_to_rgba8(t[0], t[P], t[2 * P], cc, srgb_thr);
			*(uint32_t *) (rgba + (size_t) (g.py + y) * stride_bytes + (size_t) (g.px + x) * 4) = px;
		}
		K2_PHASE(5);
		if (!BATCH) break;
		__syncthreads();
		K2_PHASE(6);
#ifdef J40_K2_PHASES
		++k2_acc[7];
#endif
	}
	K2_PHASES_END(SET - 1);   // (slots no k_vardct_dct shape uses)
}

// the three kernels: the sets' transforms need 98 / 106 / 118 registers left alone; sets 1 and 2 fit the 64 that eight wavefronts per
// SIMD leave (one register spilled / none), the AFV set gets the 96 of five (J40_K2_SPECIAL_WAVES, J40_K2_SPECIAL_WAVES_AFV)
#define J40_SPECIAL_KERNEL(NAME_, SET_, WAVES_) \
template <int NB, bool BATCH, bool XYB = false> \
__global__ void __launch_bounds__(J40_K2_SPECIAL_THREADS) __attribute__((amdgpu_waves_per_eu(WAVES_))) NAME_(DevPlan plan_arg, const DevVarblock *list, int32_t count, uint8_t *rgba, size_t stride_bytes, const K2Frame *batch, \
		const int32_t *tile_prefix, int32_t nframes, int32_t class_a, int32_t class_b) { \
	vardct_special_body<NB, BATCH, SET_, XYB>(plan_arg, list, count, rgba, stride_bytes, batch, tile_prefix, nframes, class_a, class_b); \
}
J40_SPECIAL_KERNEL(k_vardct_special_123, 1, J40_K2_SPECIAL_WAVES)
J40_SPECIAL_KERNEL(k_vardct_special_halves, 2, J40_K2_SPECIAL_WAVES)
J40_SPECIAL_KERNEL(k_vardct_special_afv, 3, J40_K2_SPECIAL_WAVES_AFV)
#undef J40_SPECIAL_KERNEL


// ------------------------------------------------------------------------------------------------
// K2l: transforms with a 128- or 256-sized side. One workgroup per varblock; the butterfly levels
// are swept over an HBM scratch (two ping-pong buffers per channel), one __syncthreads per level.
// Same arithmetic as the recursion: depth d works on sub-vectors of length N >> d (j40.h:5802-5841).

template <class Ptr>   // float * (HBM scratch) or J40_LDS float * (a panel in LDS, idct_panels)
__device__ void idct_sweeps(Ptr A, Ptr B, int32_t t, int32_t ncols, int32_t stride_k, int32_t stride_col) {
	// on return the result is in B (A is clobbered); A = input
	const int32_t N = 1 << t, half = N >> 1;
	const int32_t tid = threadIdx.x, nthreads = blockDim.x;
	if (t == 0) { for (int32_t w = tid; w < ncols; w += nthreads) B[w * stride_col] = A[w * stride_col]; __syncthreads(); return; }
	for (int32_t d = 0; d <= t - 2; ++d) {  // downward: split even / odd
		const Ptr src = (d & 1) ? B : A; const Ptr dst = (d & 1) ? A : B;
		const int32_t n = N >> d, hn = n >> 1;
		for (int32_t w = tid; w < ncols * half; w += nthreads) {
			const int32_t col = w % ncols, j = w / ncols, o = (j / hn) * n, i = j % hn;
			const Ptr s = src + col * stride_col; const Ptr q = dst + col * stride_col;
			q[(o + i) * stride_k] = s[(o + 2 * i) * stride_k];
			q[(o + hn + i) * stride_k] = i == 0 ? J40_SQRT2F * s[(o + 1) * stride_k] : s[(o + 2 * i - 1) * stride_k] + s[(o + 2 * i + 1) * stride_k];
		}
		__syncthreads();
	}
	{   // length-2 tails at depth t - 1
		const int32_t d = t - 1;
		const Ptr src = (d & 1) ? B : A; const Ptr dst = (d & 1) ? A : B;
		for (int32_t w = tid; w < ncols * half; w += nthreads) {
			const int32_t col = w % ncols, o = (w / ncols) * 2;
			const float p = src[col * stride_col + o * stride_k], q = src[col * stride_col + (o + 1) * stride_k];
			dst[col * stride_col + o * stride_k] = p + q;
			dst[col * stride_col + (o + 1) * stride_k] = p - q;
		}
		__syncthreads();
	}
	for (int32_t d = t - 2; d >= 0; --d) {  // upward: combine halves
		const Ptr src = (d & 1) ? B : A; const Ptr dst = (d & 1) ? A : B;
		const int32_t n = N >> d, hn = n >> 1;
		for (int32_t w = tid; w < ncols * half; w += nthreads) {
			const int32_t col = w % ncols, j = w / ncols, o = (j / hn) * n, i = j % hn;
			const Ptr s = src + col * stride_col; const Ptr q = dst + col * stride_col;
			const float x = s[(o + i) * stride_k], y = s[(o + hn + i) * stride_k];
			const float m = c_half_secants[hn + i];
			const float ym = y * m;
			q[(o + i) * stride_k] = x + ym;
			q[(o + n - 1 - i) * stride_k] = x - ym;
		}
		__syncthreads();
	}
}

// The same sweeps with the vectors in LDS: the 1-D transforms of `nvec` vectors of length N = 1 << t (element k of vector v at
// src[v * stride_col + k * stride_k]; one of the two strides is 1) are taken through LDS a PANEL of M vectors at a time, N * M <=
// 16384 floats, both ping-pong buffers of the sweeps in LDS (2 * (16384 + 256) floats: the panel's rows are M + 1 apart so that
// the transposing copy does not hit one bank). The result goes to dst, same layout; src is left alone. Every value takes the same
// operations in the same order as in idct_sweeps over the HBM scratch (15 levels of 256-point butterflies = 15 round trips through
// HBM per dimension there, one here).
__device__ void idct_panels(const float *src, float *dst, int32_t t, int32_t nvec, int32_t stride_k, int32_t stride_col, J40_LDS float *lds) {
	const int32_t N = 1 << t, M = min(nvec, 16384 >> t), P = M + 1;
	const int32_t tid = threadIdx.x, nthreads = blockDim.x;
	J40_LDS float *la = lds, *lb = lds + LARGE_PANEL_FLOATS;
	for (int32_t v0 = 0; v0 < nvec; v0 += M) {
		if (stride_k == 1) for (int32_t w = tid; w < N * M; w += nthreads) { const int32_t m = w >> t, k = w & (N - 1); la[k * P + m] = src[(size_t) (v0 + m) * stride_col + k]; }
		else for (int32_t w = tid; w < N * M; w += nthreads) { const int32_t k = w / M, m = w - k * M; la[k * P + m] = src[(size_t) k * stride_k + v0 + m]; }
		__syncthreads();
		idct_sweeps(la, lb, t, M, P, 1);   // result in lb
		if (stride_k == 1) for (int32_t w = tid; w < N * M; w += nthreads) { const int32_t m = w >> t, k = w & (N - 1); dst[(size_t) (v0 + m) * stride_col + k] = lb[k * P + m]; }
		else for (int32_t w = tid; w < N * M; w += nthreads) { const int32_t k = w / M, m = w - k * M; dst[(size_t) k * stride_k + v0 + m] = lb[k * P + m]; }
		__syncthreads();
	}
}

// what large_dev.h's passes run on: every lane makes the call, a barrier follows (stores to the workgroup's scratch in HBM are
// visible to its other lanes behind it)
struct WorkgroupExec {
	template <class F> __device__ __forceinline__ void run(F f) { f((int32_t) threadIdx.x, (int32_t) blockDim.x); __threadfence_block(); __syncthreads(); }
};

// REG64: the top levels of the recursion in LDS, 64-point sub-vectors in registers (large_dev.h); a tile that fits one LDS buffer
// (128x128, 128x64, 64x128) takes both dimensions there and makes ONE trip through the scratch per channel. !REG64: round 3's
// form -- every level of the butterflies as a sweep over an LDS panel, two trips per channel (J40HIP_LARGE_IDCT=sweeps).
// 512 lanes: two wavefronts per SIMD under the 133 KB of LDS -- the levels, the scatter and the colour conversion run twice as wide, the
// 256 64-point sub-vectors of a pass keep half of them busy (43 against 51 ms for the pixel stage of the maxlog-8 bench stream;
// the batch-wide instantiation then spills 1 KB per lane to scratch and is faster all the same)
#ifndef J40_LARGE_THREADS
#define J40_LARGE_THREADS 512
#endif
template <bool BATCH, bool REG64, bool XYB = false>
__global__ void __launch_bounds__(J40_LARGE_THREADS) k_vardct_large(DevPlan plan_arg, const DevVarblock *list, int32_t count, float *scratch, uint8_t *rgba, size_t stride_bytes, const K2Frame *batch, const int32_t *tile_prefix, int32_t nframes,
		int32_t class_a, int32_t class_b) {
	const int32_t tid = threadIdx.x, nthreads = blockDim.x;
	J40_STAGE_SRGB_THRESHOLDS(f);
	float *A = scratch + (size_t) blockIdx.x * 6 * 65536, *B = A + 3 * 65536;  // [3][size] each: the workgroup's own
	extern __shared__ __attribute__((aligned(16))) float large_lds[];   // 2 * LARGE_PANEL_FLOATS
	J40_LDS float *panels = (J40_LDS float *) large_lds;
	for (K2Iter it = k2_begin<BATCH>(tile_prefix, nframes); ; ) {
		int32_t frame, first; bool entered;
		if (!k2_bind<BATCH>(it, batch, tile_prefix, class_a, class_b, 1, list, count, rgba, stride_bytes, frame, first, entered)) break;
		const DevPlan &plan = BATCH ? batch[frame].plan : plan_arg;
		const DevFrame &f = *plan.frame;
		const DevVarblock vb = list[first];
		const int32_t log_rows = DEV_DCT_SELECT[vb.dctsel][0], log_columns = DEV_DCT_SELECT[vb.dctsel][1];
		const int32_t R = 1 << log_rows, C = 1 << log_columns, size = R * C;
		const ColourConsts cc = load_colour_consts(f);
		const VbGeom g = varblock_geometry(plan, vb);
		LargeSamples S;
		if constexpr (REG64) {
			WorkgroupExec ex;
			S = large_block(ex, plan, vb, g, panels, A, B, c_half_secants);
		} else {
			const int32_t long_side = R > C ? R : C, vh8 = (R < C ? R : C) / 8, vw8 = long_side / 8;
			const int32_t param_idx = vb.dctsel == 21 ? 13 : vb.dctsel <= 23 ? 14 : vb.dctsel == 24 ? 15 : 16;
			const float *dq = plan.pool_f32 + f.dq_off[param_idx];
			if (f.sparse_coeffs) {   // the tiles live in the HBM scratch here (stores are visible to the workgroup after a barrier + fence)
				for (int32_t i = tid; i < size; i += nthreads) { A[i] = 0.0f; A[65536 + i] = 0.0f; A[2 * 65536 + i] = 0.0f; }
				__threadfence_block(); __syncthreads();
				const uint16_t *order = plan.pool_u16 + f.order_off[DEV_DCT_SELECT[vb.dctsel][2] * 3];
				const TileMap map = {R, C, C, 0};
				const float qbias[3] = {f.quant_bias[0], f.quant_bias[1], f.quant_bias[2]};
				tile_scatter_events(plan, g, plan.block_events + 4 * (size_t) vb.blk, order, plan.pool_f32 + f.dq_scan_off[param_idx], size, map, A, 65536, qbias, f.quant_bias_num, tid, nthreads);
				tile_fill_llf(plan, g, long_side, vh8, vw8, map, A, 65536, f.kx_lf, f.kb_lf, tid, nthreads);
				__threadfence_block();
			} else {
				for (int32_t i = tid; i < size; i += nthreads) {
					float v[3];
					load_coeff3(plan, g, dq, size, i, long_side, vh8, vw8, v);
					const int32_t r = C > R ? i / C : i % R, c = C > R ? i % C : i / R;
					A[r * C + c] = v[0]; A[65536 + r * C + c] = v[1]; A[2 * 65536 + r * C + c] = v[2];
				}
			}
			__syncthreads();
			for (int ch = 0; ch < 3; ++ch) {
				idct_panels(A + ch * 65536, B + ch * 65536, log_columns, R, 1, C, panels);   // along c for every r: A -> B
				__threadfence_block(); __syncthreads();
				idct_panels(B + ch * 65536, A + ch * 65536, log_rows, C, C, 1, panels);      // along r for every x: B -> A
			}
			__threadfence_block(); __syncthreads();
			for (int ch = 0; ch < 3; ++ch) { S.p[ch] = A + ch * 65536; S.pitch[ch] = C; }
		}
		for (int32_t i = tid; i < size; i += nthreads) {
			const int32_t y = i >> log_columns, x = i & (C - 1);
			if (y >= g.effh || x >= g.effw) continue;
			if constexpr (XYB) { store_xyb(rgba, (size_t) (g.py + y) * stride_bytes + (size_t) (g.px + x) * 4, stride_bytes * (size_t) f.height, S.p[0][y * S.pitch[0] + x], S.p[1][y * S.pitch[1] + x], S.p[2][y * S.pitch[2] + x]); continue; }
			const uint32_t px = xyb_to_rgba8(S.p[0][y * S.pitch[0] + x], S.p[1][y * S.pitch[1] + x], S.p[2][y * S.pitch[2] + x], cc, srgb_thr);
			*(uint32_t *) (rgba + (size_t) (g.py + y) * stride_bytes + (size_t) (g.px + x) * 4) = px;
		}
		if (!BATCH) break;
		__threadfence_block(); __syncthreads();   // the next block reuses the scratch and the LDS tiles
	}
}

// ------------------------------------------------------------------------------------------------
// launch helpers

// does launch_hf_entropy take the fast path (k_hf_entropy_fast) for this frame?
bool hf_entropy_fast_path(const DevPlan &plan, const HfLaunchInfo &info) {
	static const bool allowed = [] { const char *e = getenv("J40HIP_K1_FAST"); return !e || atoi(e) != 0; }();
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	const uint32_t tables = align16(align16(info.max_num_dist) + info.max_table_bytes), wave_bytes = align16(32 * 32 * 3 + 1024 * (uint32_t) sizeof(DevGroupBlock));
	return allowed && info.lanes_fast && plan.events && info.max_clusters <= 64 && tables + HF_WAVES * wave_bytes <= 150u * 1024u;
}
// the fast path over the groups order[first .. first + count) (the caller has asked hf_entropy_fast_path)
void launch_hf_entropy_fast_ordered(const DevPlan &plan, const HfLaunchInfo &info, const uint32_t *order, int32_t first, int32_t count, hipStream_t stream) {
	if (count <= 0) return;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	const uint32_t tables = align16(align16(info.max_num_dist) + info.max_table_bytes), wave_bytes = align16(32 * 32 * 3 + 1024 * (uint32_t) sizeof(DevGroupBlock));
	static bool configured = false;
	if (!configured) { (void) hipFuncSetAttribute((const void *) k_hf_entropy_fast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
	hipLaunchKernelGGL(k_hf_entropy_fast, dim3((unsigned) ((count + HF_WAVES - 1) / HF_WAVES)), dim3(64 * HF_WAVES), tables + HF_WAVES * wave_bytes, stream, plan, first, count, tables, wave_bytes, order);
}
// block_events entries of the groups order[0 .. k): dst[i] = src[i] over each group's blocks (one workgroup per group)
__global__ void __launch_bounds__(256) k_merge_block_events(const uint32_t *group_block_start, const uint32_t *order, const uint4 *src, uint4 *dst) {
	const uint32_t g = order[blockIdx.x], b0 = group_block_start[g], b1 = group_block_start[g + 1];
	for (uint32_t i = b0 + threadIdx.x; i < b1; i += 256) dst[i] = src[i];
}
// the rectangles of the groups order[0 .. k) of the device image `src` written into the (pinned, device-visible) host image `dst`:
// workgroup (row, i) copies one row of group i's rectangle, a dword per lane and turn
__global__ void __launch_bounds__(256) k_store_group_rects(const uint32_t *order, int32_t gcolumns, int32_t shift, int32_t width, int32_t height, const uint8_t *src, uint8_t *dst, size_t stride_bytes) {
	const uint32_t g = order[blockIdx.y];
	const int32_t x0 = (int32_t) (g % (uint32_t) gcolumns) << shift, y = ((int32_t) (g / (uint32_t) gcolumns) << shift) + (int32_t) blockIdx.x;
	if (y >= height) return;
	const int32_t w = min(1 << shift, width - x0);
	const size_t off = (size_t) y * stride_bytes + (size_t) x0 * 4;
	const uint32_t *s = (const uint32_t *) (src + off);
	uint32_t *d = (uint32_t *) (dst + off);
	for (int32_t x = threadIdx.x; x < w; x += 256) d[x] = s[x];
}
void launch_store_group_rects(const uint32_t *order, int32_t k, int32_t gcolumns, int32_t shift, int32_t width, int32_t height, const uint8_t *src, uint8_t *dst_host_mapped, size_t stride_bytes, hipStream_t stream) {
	if (k > 0) hipLaunchKernelGGL(k_store_group_rects, dim3(1u << shift, (unsigned) k), dim3(256), 0, stream, order, gcolumns, shift, width, height, src, dst_host_mapped, stride_bytes);
}
void launch_merge_block_events(const DevPlan &plan, const uint32_t *order, int32_t k, const uint32_t *shadow, hipStream_t stream) {
	if (k > 0) hipLaunchKernelGGL(k_merge_block_events, dim3((unsigned) k), dim3(256), 0, stream, plan.group_block_start, order, (const uint4 *) shadow, (uint4 *) plan.block_events);
}

void launch_hf_entropy(const DevPlan &plan, const HfLaunchInfo &info, int32_t first_group, int32_t num_groups, hipStream_t stream) {
	if (num_groups <= 0) return;
	HfLdsLayout lay;
	auto align16 = [](uint32_t v) { return (v + 15u) & ~15u; };
	{
		// single-pass frames (sparse coefficients) with the throughput kernel's kind of tables: the fast path (J40HIP_K1_FAST=0: never)
		static const bool allowed = [] { const char *e = getenv("J40HIP_K1_FAST"); return !e || atoi(e) != 0; }();
		const uint32_t tables = align16(align16(info.max_num_dist) + info.max_table_bytes), wave_bytes = align16(32 * 32 * 3 + 1024 * (uint32_t) sizeof(DevGroupBlock));
		if (allowed && info.lanes_fast && plan.events && info.max_clusters <= 64 && tables + HF_WAVES * wave_bytes <= 150u * 1024u) {
			static bool configured = false;
			if (!configured) { (void) hipFuncSetAttribute((const void *) k_hf_entropy_fast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
			hipLaunchKernelGGL(k_hf_entropy_fast, dim3((unsigned) ((num_groups + HF_WAVES - 1) / HF_WAVES)), dim3(64 * HF_WAVES), tables + HF_WAVES * wave_bytes, stream, plan, first_group, num_groups, tables, wave_bytes, (const uint32_t *) nullptr);
			return;
		}
	}
	uint32_t off = 0;
	lay.off_bctx = off; off = align16(off + info.block_ctx_size);
	lay.off_nnz = off; off += 128;
	lay.off_freq = off; off += 64;
	const bool in_lds = info.tables_fit_lds;
	lay.off_map = off; off = align16(off + (in_lds ? info.max_num_dist : 0));
	lay.off_clusters = off; off = align16(off + (in_lds ? info.max_clusters * (uint32_t) sizeof(DevCluster) : 0));
	lay.off_tables = off; off = align16(off + (in_lds ? info.max_table_bytes : 0));
	lay.off_wave = off;
	lay.off_wave_blocks = 32 * 32 * 3;
	lay.wave_bytes = align16(lay.off_wave_blocks + 1024 * (uint32_t) sizeof(DevGroupBlock));
	lay.total = lay.off_wave + HF_WAVES * lay.wave_bytes;
	const unsigned blocks = (unsigned) ((num_groups + HF_WAVES - 1) / HF_WAVES);
	if (in_lds) {
		static bool configured = false;
		if (!configured) { (void) hipFuncSetAttribute((const void *) k_hf_entropy<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); configured = true; }
		hipLaunchKernelGGL(k_hf_entropy<true>, dim3(blocks), dim3(64 * HF_WAVES), lay.total, stream, plan, first_group, num_groups, lay);
	} else {
		hipLaunchKernelGGL(k_hf_entropy<false>, dim3(blocks), dim3(64 * HF_WAVES), lay.total, stream, plan, first_group, num_groups, lay);
	}
}

// `batch` = nullptr: one frame, everything in the kernel arguments. Otherwise a persistent launch over the tiles of one class of
// `nframes` frames (k2_bind): `grid` workgroups, tile_prefix = this launch's row of the table k_k2_tiles built; plan / list /
// count / rgba / stride are then ignored, `large_scratch` holds 6 * 65536 floats per workgroup of the launch.
struct K2Launch { const K2Frame *batch; const int32_t *tile_prefix; int32_t nframes, class_a, class_b, grid; int32_t xyb; };   // xyb: single-frame launches, store_xyb instead of the colour tail

template <int LOGR, int LOGC, int NB>
static void launch_dct(const DevPlan &plan, const DevVarblock *list, int32_t count, int32_t param_idx, int32_t order_idx, uint8_t *rgba, size_t stride, const K2Launch &bl, hipStream_t stream) {
	constexpr size_t lds_bytes = (size_t) NB * 3 * (1 << LOGR) * ((1 << LOGC) + 1) * sizeof(float);
	static bool configured = false;
	if (!configured) {
		(void) hipFuncSetAttribute((const void *) k_vardct_dct<LOGR, LOGC, NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
		(void) hipFuncSetAttribute((const void *) k_vardct_dct<LOGR, LOGC, NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
		(void) hipFuncSetAttribute((const void *) k_vardct_dct<LOGR, LOGC, NB, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
		configured = true;
	}
	const int32_t blocks = (count + NB - 1) / NB;
	if (!bl.batch && bl.xyb) hipLaunchKernelGGL((k_vardct_dct<LOGR, LOGC, NB, false, true>), dim3((unsigned) blocks), dim3(256), lds_bytes, stream, plan, list, count, param_idx, order_idx, rgba, stride, bl.batch, nullptr, 1, 0, 0);
	else if (bl.batch) hipLaunchKernelGGL((k_vardct_dct<LOGR, LOGC, NB, true>), dim3((unsigned) bl.grid), dim3(256), lds_bytes, stream, plan, list, count, param_idx, order_idx, rgba, stride, bl.batch, bl.tile_prefix, bl.nframes, bl.class_a, bl.class_b);
	else hipLaunchKernelGGL((k_vardct_dct<LOGR, LOGC, NB, false>), dim3((unsigned) blocks), dim3(256), lds_bytes, stream, plan, list, count, param_idx, order_idx, rgba, stride, bl.batch, nullptr, 1, 0, 0);
}

// J40HIP_K2_WIDE: bit 0: the batched 8x8 DCT takes 32 blocks per workgroup instead of 16, bit 1: 16x8 / 8x16 take 16 instead of 8 (experiments)
static int k2_wide() { static const int v = [] { const char *e = getenv("J40HIP_K2_WIDE"); return e ? atoi(e) : 0; }(); return v; }

// list = varblocks of one DctSelect value (of a run of values for the 8x8 specials and for the 128/256-sized transforms)
static void launch_vardct_class_impl(const DevPlan &plan, int32_t dctsel, const DevVarblock *list, int32_t count, float *large_scratch, uint8_t *rgba, size_t stride, const K2Launch &bl, hipStream_t stream) {
	if (count <= 0 && !bl.batch) return;
	switch (dctsel) {
	case 0: if (bl.batch && (k2_wide() & 1)) launch_dct<3, 3, 32>(plan, list, count, 0, 0, rgba, stride, bl, stream); else launch_dct<3, 3, 16>(plan, list, count, 0, 0, rgba, stride, bl, stream); break;
	case 4: launch_dct<4, 4, 8>(plan, list, count, 4, 2, rgba, stride, bl, stream); break;
	case 5: launch_dct<5, 5, 2>(plan, list, count, 5, 3, rgba, stride, bl, stream); break;
	case 6: if (bl.batch && (k2_wide() & 2)) launch_dct<4, 3, 16>(plan, list, count, 6, 4, rgba, stride, bl, stream); else launch_dct<4, 3, 8>(plan, list, count, 6, 4, rgba, stride, bl, stream); break;
	case 7: if (bl.batch && (k2_wide() & 2)) launch_dct<3, 4, 16>(plan, list, count, 6, 4, rgba, stride, bl, stream); else launch_dct<3, 4, 8>(plan, list, count, 6, 4, rgba, stride, bl, stream); break;
	case 8: launch_dct<5, 3, 4>(plan, list, count, 7, 5, rgba, stride, bl, stream); break;
	case 9: launch_dct<3, 5, 4>(plan, list, count, 7, 5, rgba, stride, bl, stream); break;
	case 10: launch_dct<5, 4, 4>(plan, list, count, 8, 6, rgba, stride, bl, stream); break;
	case 11: launch_dct<4, 5, 4>(plan, list, count, 8, 6, rgba, stride, bl, stream); break;
	case 18: launch_dct<6, 6, 1>(plan, list, count, 11, 7, rgba, stride, bl, stream); break;
	case 19: launch_dct<6, 5, 1>(plan, list, count, 12, 8, rgba, stride, bl, stream); break;
	case 20: launch_dct<5, 6, 1>(plan, list, count, 12, 8, rgba, stride, bl, stream); break;
#define J40_LAUNCH_SPECIAL(KERNEL_) do { \
		if (!bl.batch && bl.xyb) hipLaunchKernelGGL((KERNEL_<J40_K2_SPECIAL_NB, false, true>), dim3((unsigned) ((count + J40_K2_SPECIAL_NB - 1) / J40_K2_SPECIAL_NB)), dim3(J40_K2_SPECIAL_THREADS), 0, stream, plan, list, count, rgba, stride, bl.batch, nullptr, 1, 0, 0); \
		else if (bl.batch) hipLaunchKernelGGL((KERNEL_<J40_K2_SPECIAL_NB, true>), dim3((unsigned) bl.grid), dim3(J40_K2_SPECIAL_THREADS), 0, stream, plan, list, count, rgba, stride, bl.batch, bl.tile_prefix, bl.nframes, bl.class_a, bl.class_b); \
		else hipLaunchKernelGGL((KERNEL_<J40_K2_SPECIAL_NB, false>), dim3((unsigned) ((count + J40_K2_SPECIAL_NB - 1) / J40_K2_SPECIAL_NB)), dim3(J40_K2_SPECIAL_THREADS), 0, stream, plan, list, count, rgba, stride, bl.batch, nullptr, 1, 0, 0); \
	} while (0)
	case 1: case 2: case 3: J40_LAUNCH_SPECIAL(k_vardct_special_123); break;
	case 12: case 13: J40_LAUNCH_SPECIAL(k_vardct_special_halves); break;
	case 14: case 15: case 16: case 17: J40_LAUNCH_SPECIAL(k_vardct_special_afv); break;
#undef J40_LAUNCH_SPECIAL
	default:
		{
			constexpr size_t lds_bytes = 2 * (size_t) LARGE_PANEL_FLOATS * sizeof(float);
			static bool configured = false;
			if (!configured) {
				(void) hipFuncSetAttribute((const void *) k_vardct_large<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
				(void) hipFuncSetAttribute((const void *) k_vardct_large<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
				(void) hipFuncSetAttribute((const void *) k_vardct_large<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
				(void) hipFuncSetAttribute((const void *) k_vardct_large<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
				(void) hipFuncSetAttribute((const void *) k_vardct_large<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes);
				configured = true;
			}
			// (J40HIP_LARGE_IDCT=sweeps: round 3's kernel, every butterfly level as a sweep over an LDS panel -- kept for comparison)
			static const bool reg64 = [] { const char *e = getenv("J40HIP_LARGE_IDCT"); return !(e && !strcmp(e, "sweeps")); }();
			if (bl.batch) {
				if (reg64) hipLaunchKernelGGL((k_vardct_large<true, true>), dim3((unsigned) bl.grid), dim3(J40_LARGE_THREADS), lds_bytes, stream, plan, list, count, large_scratch, rgba, stride, bl.batch, bl.tile_prefix, bl.nframes, bl.class_a, bl.class_b);
				else hipLaunchKernelGGL((k_vardct_large<true, false>), dim3((unsigned) bl.grid), dim3(J40_LARGE_THREADS), lds_bytes, stream, plan, list, count, large_scratch, rgba, stride, bl.batch, bl.tile_prefix, bl.nframes, bl.class_a, bl.class_b);
			} else if (bl.xyb) {
				hipLaunchKernelGGL((k_vardct_large<false, true, true>), dim3((unsigned) count), dim3(J40_LARGE_THREADS), lds_bytes, stream, plan, list, count, large_scratch, rgba, stride, bl.batch, nullptr, 1, 0, 0);
			} else {
				if (reg64) hipLaunchKernelGGL((k_vardct_large<false, true>), dim3((unsigned) count), dim3(J40_LARGE_THREADS), lds_bytes, stream, plan, list, count, large_scratch, rgba, stride, bl.batch, nullptr, 1, 0, 0);
				else hipLaunchKernelGGL((k_vardct_large<false, false>), dim3((unsigned) count), dim3(J40_LARGE_THREADS), lds_bytes, stream, plan, list, count, large_scratch, rgba, stride, bl.batch, nullptr, 1, 0, 0);
			}
		}
		break;
	}
}
void launch_vardct_class(const DevPlan &plan, int32_t dctsel, const DevVarblock *list, int32_t count, float *large_scratch, uint8_t *rgba, size_t stride, hipStream_t stream) {
	launch_vardct_class_impl(plan, dctsel, list, count, large_scratch, rgba, stride, K2Launch{nullptr, nullptr, 1, 0, 0, 0, 0}, stream);
}

// known-answer hook: the renderer's per-sample tail (sRGB transfer + conversion, j40.h:7213-7240 / 7925-7935)
__global__ void k_kat_srgb_u8(const float *v, size_t n, uint8_t *out) {
	__shared__ float s_thr[SRGB_TABLE_FLOATS];
	for (int32_t i = threadIdx.x; i < SRGB_TABLE_FLOATS; i += blockDim.x) s_thr[i] = c_srgb_thr[i];
	__syncthreads();
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float x = v[i];
	int32_t px;
	if (x > -9.0f && x < 50000.0f) px = srgb_u8_from_thresholds(x, (const J40_LDS float *) s_thr);   // as xyb_to_rgba8 does for 8-bit frames
	else { px = f32_to_i16_x86(255.0f * srgb_transfer(x) + 0.5f); px = px < 0 ? 0 : px > 255 ? 255 : px; }
	out[i] = (uint8_t) px;
}
void launch_kat_srgb_u8(const float *v, size_t n, uint8_t *out, hipStream_t stream) {
	hipLaunchKernelGGL(k_kat_srgb_u8, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, v, n, out);
}

// every class of one frame; the nine 8x8 special transforms (DctSelect 1-3, 12-13 and 14-17, each run contiguous in the
// sorted list) share three launches: k_vardct_special dispatches per varblock among the transforms of its set
static bool class_range(int d, int *a, int *b) {
	if (d == 2 || d == 3 || d == 13 || (d >= 15 && d <= 17)) return false;
	*a = d; *b = d == 1 ? 4 : d == 12 ? 14 : d == 14 ? 18 : d + 1;
	return true;
}
void launch_vardct_frame(const DevPlan &plan, const int32_t *class_start, const DevVarblock *sorted, float *large_scratch, uint8_t *rgba, size_t stride, hipStream_t stream) {
	for (int d = 0, a, b; d < 27; ++d) if (class_range(d, &a, &b))
		launch_vardct_class(plan, d, sorted + class_start[a], class_start[b] - class_start[a], large_scratch, rgba, stride, stream);
}
// the same with the samples left in XYB: three float planes of the frame from `xyb`, `stride` bytes per row (store_xyb)
void launch_vardct_frame_xyb(const DevPlan &plan, const int32_t *class_start, const DevVarblock *sorted, float *large_scratch, float *xyb, size_t stride, hipStream_t stream) {
	for (int d = 0, a, b; d < 27; ++d) if (class_range(d, &a, &b))
		launch_vardct_class_impl(plan, d, sorted + class_start[a], class_start[b] - class_start[a], large_scratch, (uint8_t *) xyb, stride, K2Launch{nullptr, nullptr, 1, 0, 0, 0, 1}, stream);
}

// ---- the same for every frame of a batch at once: one persistent launch per class ----
// The launches of a batch: {first DctSelect, one past the last, varblocks per tile, cells per varblock at least}; the 128/256-sized
// transforms share one (k_vardct_large looks at each block's DctSelect). Biggest first.
#define J40_K2_LAUNCH_TABLE {0, 1, 16, 1}, {1, 4, J40_K2_SPECIAL_NB, 1}, {12, 14, J40_K2_SPECIAL_NB, 1}, {14, 18, J40_K2_SPECIAL_NB, 1}, {4, 5, 8, 4}, {6, 7, 8, 2}, {7, 8, 8, 2}, {5, 6, 2, 16}, {8, 9, 4, 4}, {9, 10, 4, 4}, {10, 11, 4, 8}, {11, 12, 4, 8}, \
	{18, 19, 1, 64}, {19, 20, 1, 32}, {20, 21, 1, 32}, {21, 27, 1, 128}
struct K2BatchLaunch { int16_t a, b, per_wg, min_cells; };
struct K2Table { K2BatchLaunch l[K2_NUM_BATCH_LAUNCHES]; };
static const K2Table &k2_table() {
	static const K2Table t = [] { K2Table t = {{J40_K2_LAUNCH_TABLE}}; if (k2_wide() & 1) t.l[0].per_wg = 32; if (k2_wide() & 2) t.l[5].per_wg = t.l[6].per_wg = 16; return t; }();
	return t;
}

// tile_prefix[l * (nframes + 1) + f] = tiles of launch l in the frames before f; totals[l] = all of them; one thread per launch
__global__ void k_k2_tiles(const K2Frame *frames, int32_t nframes, int32_t *tile_prefix, int32_t *totals, K2Table table) {
	const int32_t l = threadIdx.x;
	if (l >= K2_NUM_BATCH_LAUNCHES) return;
	const int32_t a = table.l[l].a, b = table.l[l].b, per = table.l[l].per_wg;
	int32_t at = 0;
	int32_t *row = tile_prefix + l * (nframes + 1);
	for (int32_t f = 0; f < nframes; ++f) {
		row[f] = at;
		const int32_t n = frames[f].class_start[b] - frames[f].class_start[a];
		at += (n + per - 1) / per;
	}
	row[nframes] = at;
	totals[l] = at;
}

// Workgroups per launch, and which of (up to) four side streams a launch goes to. A process gets four hardware queues per
// priority (GPU_MAX_HW_QUEUES); streams beyond that share them, and kernels in one queue run one after the other. So the sixteen
// launches are dealt to four streams as four chains of about equal work (measured shares of the picture-encoded 8K stream: the 8x8
// DCT 18 %, the 8x8 specials 38 %, 16x16 11 %, ...), every launch wide enough to fill the machine on its own: whichever chains are
// still running share it, and a chain's last kernel has the whole machine for its tail. (Fifteen streams with grids in proportion
// to the classes' work were measured: 176 ms per 256 frames against 81 -- only four kernels ran at a time, each with a fraction of
// the machine.)
// (the second launch of the specials goes behind the 8x8 DCT: beside the first it made its chain the longest by 15 ms, alone at the
// end with two workgroups per compute unit. J40HIP_K2_CHAINS=<16 digits>: another assignment, for experiments)
static const int8_t *k2_launch_stream() {
	static int8_t chain[K2_NUM_BATCH_LAUNCHES] = {0, 1, 0, 1, 2, 3, 3, 2, 3, 3, 3, 3, 2, 2, 2, 2};
	static const bool once = [] { const char *e = getenv("J40HIP_K2_CHAINS"); if (e && strlen(e) == K2_NUM_BATCH_LAUNCHES) for (int i = 0; i < K2_NUM_BATCH_LAUNCHES; ++i) chain[i] = (int8_t) ((e[i] - '0') & 3); return true; }();
	(void) once;
	return chain;
}
void k2_batch_grids(const int32_t *last_totals, size_t cells_total, int32_t nframes, int32_t wg_slots, int32_t *grids) {
	for (int l = 0; l < K2_NUM_BATCH_LAUNCHES; ++l) {
		const auto &L = k2_table().l[l];
		const size_t bound = last_totals ? (size_t) last_totals[l] + (size_t) last_totals[l] / 4 + 8 : cells_total / (size_t) (L.min_cells * L.per_wg) + (size_t) nframes;   // tiles, about
		int64_t g = std::min<int64_t>((int64_t) bound, wg_slots);
		// (the 128/256-sized transforms' workgroups want 133 KB of LDS each: a launch waits for compute units to drain even when it
		// has nothing to do -- 11 ms at the end of its chain with 256 workgroups -- so a batch following one without such blocks gets one)
		if (L.a == 21) g = last_totals && last_totals[l] == 0 ? 1 : std::min<int64_t>(g, K2_LARGE_WGS);
		grids[l] = (int32_t) std::max<int64_t>(g, 1);
	}
}

// grids: workgroups per launch (k2_batch_grids); large_scratch: 6 * 65536 floats per workgroup of the last launch (K2_LARGE_WGS of
// them); totals_dev: K2_NUM_BATCH_LAUNCHES ints, the tiles each launch found
void launch_vardct_batch(const K2Frame *frames_dev, int32_t nframes, int32_t *tile_prefix_dev, int32_t *totals_dev, const int32_t *grids, float *large_scratch, hipStream_t stream, hipStream_t *side, int nside, hipEvent_t fork, hipEvent_t *side_done) {
	const DevPlan none = DevPlan();
	hipLaunchKernelGGL(k_k2_tiles, dim3(1), dim3(32), 0, stream, frames_dev, nframes, tile_prefix_dev, totals_dev, k2_table());
	if (nside > 0) { (void) hipEventRecord(fork, stream); for (int k = 0; k < nside; ++k) (void) hipStreamWaitEvent(side[k], fork, 0); }
	for (int l = 0; l < K2_NUM_BATCH_LAUNCHES; ++l) {
		const auto &L = k2_table().l[l];
		launch_vardct_class_impl(none, L.a, nullptr, 0, large_scratch, nullptr, 0, K2Launch{frames_dev, tile_prefix_dev + (size_t) l * (size_t) (nframes + 1), nframes, L.a, L.b, grids[l], 0}, nside > 0 ? side[k2_launch_stream()[l] % nside] : stream);
	}
	if (nside > 0) for (int k = 0; k < nside; ++k) { (void) hipEventRecord(side_done[k], side[k]); (void) hipStreamWaitEvent(stream, side_done[k], 0); }
}

// ------------------------------------------------------------------------------------------------
// the restoration filters (they share this unit's constant tables)
#include "restore_kernels.h"

} // namespace j40hip
