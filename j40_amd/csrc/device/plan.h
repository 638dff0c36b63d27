// j40_amd/csrc/device/plan.h -- the frame plan as the HIP kernels see it: flat, pointer-free PODs
// mirroring what the reference's hot path reads from j40__frame_st / j40__lf_group_st / j40__code_spec
// (j40.h:5061-5122, 6360-6390, 2486-2495). Offsets index into a handful of typed pools in HBM.
#pragma once
#include <stdint.h>

namespace j40hip {

struct DevCluster {
	uint32_t cfg;        // hybrid-int config: bits 0-3 split_exp, 4-7 msb_in_token, 8-11 lsb_in_token
	int32_t max_token;
	uint32_t table_off;  // ANS: first AnsEntry of this cluster in the u64 pool; prefix: first int32 in the i32 pool
	int16_t fast_len, max_len;  // prefix codes only
};

struct DevCodeSpec {
	int32_t num_dist, num_clusters;
	int32_t lz77_enabled, use_prefix_code;
	int32_t min_symbol, min_length;
	int32_t log_alpha_size;
	uint32_t lz_len_cfg;
	int32_t lz_len_max_token;
	uint32_t cluster_map_off;  // into the u8 pool
	uint32_t cluster_off;      // into the DevCluster pool
	uint32_t table_span;       // elements this spec's alias (u64) / prefix (i32) tables occupy in their pool, contiguous
	// throughput-form K1 (hf_lanes_dev.h), rANS specs without LZ77 and with uniformly sized alias tables only, else
	// 0xffffffff: one word per cluster in the i32 pool, hybrid config | min(max_token, 0xfffff) << 12
	uint32_t lane_cfg_off;
};

struct DevLfGroup {
	int32_t left, top, width, height;
	int32_t width8, height8, width64, height64;
	int32_t cell_base;   // first 8x8 cell of this LF group in the frame-wide cell arrays
	int32_t vb_base;     // first varblock in the frame-wide varblock arrays
	int32_t c64_base;    // first 64x64 cell (chroma-from-luma factors)
	int32_t nb_varblocks;
	float mult_lf[3];    // LF dequantisation factors of this group, channels X, Y, B (j40.h:6562; they depend on the group's extra_prec)
	int32_t pad;
};

// one (pass, group) TOC section
struct DevSection {
	uint32_t byte_off, size, bit_off;
	int32_t ggidx;
	int32_t gx8, gy8;    // cell offset of the group inside its LF group
	int32_t gw8, gh8;    // cells
	int32_t gx, gy, gw, gh;  // pixels: group position in the frame and size
};

// K1 work list: the varblocks of one group in the order j40__hf_coeffs visits them (raster order of
// their top-left cells, j40.h:6907-6915), with everything the context model needs
struct DevGroupBlock {
	uint32_t coeffoff_qfidx;  // as j40__varblock (j40.h:6352): coefficient offset | qf index
	uint16_t pos_dct;         // bits 0-9: y8 * 32 + x8 inside the group; bits 10-14: DctSelect
	uint16_t bctx3;           // block context (j40.h:6951-6953, < 16) of channel Y | X << 4 | B << 8, looked up on the host
};

// work item of the coefficients -> pixels kernels, sorted by DctSelect on the host. Self-contained: everything the kernels
// need to place and scale the block is resolved on the host, so a workgroup starts with one 32-byte read per block instead of
// a chain of dependent loads (list -> varblock arrays -> LF group -> chroma-from-luma cell).
struct DevVarblock {
	int32_t coeff_base;   // index of the block's first coefficient in plan.coeffs[c]
	int32_t llf_base;     // index of the block's first LLF coefficient in plan.llf[c]
	float mult1;          // 65536 / global_scale / HfMul (j40.h:7078-7080), the Y channel's multiplier
	float kx_hf;          // chroma-from-luma factors of the block's 64x64 cell: base_corr + inv_colour_factor * x/bfromy (j40.h:7138-7143)
	int32_t px, py;       // top-left pixel in the frame
	uint16_t effw, effh;  // visible size
	uint8_t dctsel, pad[3];   // pad: the block's LF group index, little endian (lf_tail_kernels.hip)
	int32_t blk;          // ordinal of the block in plan.group_blocks / plan.block_events
	float kb_hf;
};

struct DevFrame {
	int32_t width, height;
	int32_t num_passes, num_groups, num_lf_groups;
	int32_t nb_block_ctx, nb_qf_thr, lfidx_size, num_hf_presets, preset_bits;
	int32_t bpp;
	float quant_bias[3], quant_bias_num;
	float mult_base;             // 65536.0f / (float) global_scale (j40.h:7078)
	float x_qm_mul, b_qm_mul;    // 0.8^(qm_scale - 2) (j40.h:7055)
	float kx_lf, kb_lf;          // j40.h:7115-7116
	float base_corr_x, base_corr_b, inv_colour_factor;
	float opsin_inv_mat[9], opsin_bias[3], cbrt_opsin_bias[3];
	float itscale;               // 255.0f / intensity_target
	uint32_t order_off[11 * 13 * 3];  // into the u16 pool; 0xffffffff = not loaded
	uint32_t dq_off[17];              // into the f32 pool, layout [channel][coefficient]; 0xffffffff = not loaded
	uint32_t dq_size[17];
	// single-pass frames: the same weights gathered through pass 0's coefficient order, dq_scan[c][pos] = dq[c][order[c][pos]], so
	// that an event's weight is found from its scan position without waiting for the order lookup; 0xffffffff = not built
	uint32_t dq_scan_off[17];
	// single-pass frames: the entropy kernel does not fill dense coefficient planes; it appends (scan position, value)
	// events per block, see DevPlan::events. 0: dense planes in canonical order, accumulated over the passes (j40.h:6989)
	int32_t sparse_coeffs;
	// frames with extra channels: a Modular sub-image follows the HF coefficients in every pass-group section (j40.h:7024-7034).
	// The reference decodes it and then drops it (j40__combine_vardct replaces the channel list, j40.h:7868-7870: the output is
	// opaque); here it is not decoded, so a section does not have to end where its coefficients end.
	int32_t sections_have_trailer;           // pass 0: the three channels share one coefficient order (the usual case)
	// Whether the end of a section is checked. Only in frames that are one section (then: zero padding to the byte boundary,
	// j40.h:8203, and bytes left over are `shrt`, j40.h:7796-7803). In frames with several sections the reference checks
	// nothing: j40__finish_section_state (j40.h:7778-7795) runs j40__no_more_bytes on the section's own state and returns the
	// parent's error code, so `pad0` / `excs` never surface -- junk behind a section's data is accepted, and so it is here.
	int32_t check_section_end;
	uint32_t single_declared_end;            // single-section frames: where the TOC says the section ends (byte offset); it is readable to the
	                                         // end of the codestream, stopping short of this is `shrt`, going past it `excs` (j40.h:7796-7803)
};

// one non-zero quantised HF coefficient in FOUR bytes: scan position inside its block (a block has at most 256 x 256 positions) in the
// low half, the value as int16 in the high half. A coefficient beyond 16 bits -- legal, never seen -- makes its section report ERR_EVOF
// like a full event region does, and the frame is decoded with dense planes. (Eight-byte events until round 3: the entropy kernel
// wrote 27 GB of them per 256 8K frames and the pixel kernels read them back; half of that now, and 72 MB less per frame in flight.)
struct CoeffEvent { uint32_t packed; };
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t coeff_event_pack(uint32_t pos, int32_t value) { return (pos & 0xffffu) | ((uint32_t) value << 16); }
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t coeff_event_pos(CoeffEvent e) { return e.packed & 0xffffu; }
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
int32_t coeff_event_value(CoeffEvent e) { return (int32_t) (int16_t) (e.packed >> 16); }
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
bool coeff_event_fits(int32_t value) { return value >= -32768 && value <= 32767; }

// everything a kernel needs, passed by value
struct DevPlan {
	const DevFrame *frame;
	const uint8_t *codestream;
	const uint8_t *pool_u8;          // cluster maps, block context map
	const uint16_t *pool_u16;        // coefficient orders
	const int32_t *pool_i32;         // prefix code tables
	const uint64_t *pool_u64;        // rANS alias tables
	const float *pool_f32;           // dequantisation weights
	const DevCluster *clusters;
	const DevCodeSpec *coeff_specs;  // [num_passes]
	const DevLfGroup *lf_groups;
	const DevSection *sections;      // [num_passes * num_groups]
	const DevGroupBlock *group_blocks;   // concatenated per group
	const uint32_t *group_block_start;   // [num_groups + 1]
	uint32_t block_ctx_map_off;      // into pool_u8
	// LF bundle, frame-wide arrays concatenated over LF groups
	const int32_t *blocks;
	const uint8_t *lfindices;
	const float *llf[3];
	const int32_t *vb_coeffoff_qfidx;
	const float *vb_hfmul_inv;
	const int16_t *xfromy, *bfromy;
	// frames whose LfGroup tail runs on the device (Frame::defer_lf_tail): the decoded LF integers, frame-wide cell arrays like `blocks`,
	// channels X, Y, B; lf_tail_kernels.hip turns them into `llf` at upload. Null otherwise (then the host filled `llf`).
	const int16_t *lfraw[3];
	// working buffers
	float *coeffs[3];                // [total cells * 64]; one allocation: coeffs[c] = coeffs[0] + c * coeff_stride
	uint32_t coeff_stride;
	// Sparse coefficients (DevFrame::sparse_coeffs). At typical qualities two thirds of the quantised HF coefficients are zero,
	// and a dense plane costs 12 B/pixel to clear, to fill with scattered 4-byte stores and to read back. Instead the entropy
	// kernel appends one event {scan position, value} per non-zero coefficient to its section's region of `events` -- sequential
	// 8-byte stores -- and notes per block where its events start and how many each channel has (emission order Y, X, B:
	// block_events[4 * blk] = first event, [+1..+3] = counts). The pixel kernels zero a tile in LDS and scatter the events
	// into it, so dequantisation work is proportional to the non-zeros. ev_range[2 * g], [2 * g + 1]: first / end event
	// index of group g's region (sized on the host from the section's byte size; running out of it is ERR_EVOF).
	CoeffEvent *events;
	const uint32_t *ev_range;
	uint32_t *block_events;
	int8_t *nonzeros;                // [num_groups][32 * 32 * 3]
	int32_t *lz_window;              // [num_groups][lz_window_size] or null
	uint32_t lz_window_size;
	uint32_t *status;                // [num_passes * num_groups] 4-char codes
	uint32_t *section_end_bit;       // [num_passes * num_groups] or null: where each section's HF coefficients ended (absolute bit),
	                                 // written by the latency-form entropy kernel of frames whose sections go on with a Modular sub-image
	const uint32_t *lane_order;      // [num_groups] or null: the group k_hf_lanes gives its k-th lane -- the groups by decreasing section size, so
	                                 // that the 64 lanes of a wavefront decode sections of about the same length (null: group k)
};

// ---- Modular frames ----

// MA tree node, same 16-byte layout as the host's TreeNode (modular.hpp):
//   branch: prop >= 0, value = threshold, a / b = relative offsets of the (> threshold) / (<=) child
//   leaf:   prop = -1 - predictor, value = context, a = offset, b = multiplier
struct DevTreeNode { int32_t prop, value, a, b; };

// one pass-group section of a Modular frame, header already parsed on the host
struct DevModSection {
	uint32_t byte_off, size, bit_off;   // bit_off: where the channel residuals start
	int32_t gx, gy, gw, gh;             // group rectangle in the frame
	int32_t sidx;                       // stream index property (j40.h:7013; 0 for LfGlobal)
	int32_t first_channel, num_channels;  // channels of the global image this section codes
	int8_t wp[12];                      // weighted predictor parameters p1, p2, p3[5], w[4] (j40.h:3551)
	// the MA tree and code spec this section decodes with: the global ones, or its own (use_global_tree = 0, j40.h:3740)
	uint32_t tree_off;                  // first node in DevModPlan::tree
	int32_t tree_nodes, spec_idx, uses_wp;
	// reversible colour transforms listed in the section's own header (j40.h:3757), undone over the section's
	// rectangle after its channels are decoded (j40.h:7030): pairs {begin_c, rct_type} at DevModPlan::local_rct + 2 * local_off
	int32_t local_off, local_count;
	// a section whose own header lists a palette decodes into planes of its own (DevModPlan::sub_planes[sub_off ..], one per coded
	// channel, tightly packed) -- the sub-image the reference allocates (j40.h:7024-7031); the host then schedules its inverse
	// transforms and pastes the result into the frame planes. -1: the channels are the frame planes first_channel ...
	int32_t sub_off;
	// != 0: the section's own Modular header did not parse (on the host); nothing is decoded and this becomes the section's status,
	// so that it takes its place among the other sections' errors (the first failing section in stream order is reported)
	uint32_t preset_status;
	// frames whose channels differ in size (Squeeze): the section's channels as explicit rectangles,
	// DevModPlan::chan_rects[chan_off .. chan_off + num_channels). -1: first_channel ... over the rectangle above
	int32_t chan_off;
	// >= 0: the section is decoded by the wave-cooperative kernel (modular_coop.hip) with DevModPlan::coop_trees[coop_idx];
	// -1: by k_modular_sections
	int32_t coop_idx;
	// LZ77 distance multiplier of the section's stream + 1 (j40.h:3840-3844), or 0: the widest non-meta channel among the section's own
	// channels (a pass group's sub-image, j40.h:7024). LfGlobal's section belongs to the frame-wide image: its multiplier comes from
	// ALL of that image's non-meta channels, also the ones the section itself does not code (multi-group frames: only the palette is)
	int32_t dist_mult_p1;
	// != 0 (only with coop_idx >= 0): four such sections share a wavefront of k_modular_quad (modular_quad.hip); frames with
	// thousands of sections
	int32_t quad;
	// != 0: the section's MA tree looks only at where a sample is (properties 0-3, no weighted predictor): decoded in two passes by
	// modular_split.hip -- its residual tokens at DevModPlan::residuals + res_off (res_count of them, stream order). 2: the tree tests
	// the column (property 3), so the leaf is looked up per sample, not per row
	int32_t split;
	uint32_t res_off, res_count;
};

// An MA tree laid out for a wavefront that decodes ONE section with all 64 lanes (k_modular_coop): lane i holds branch node i
// (property, threshold) and leaf i. Per sample every lane evaluates its node's test at once (one ballot = the outcome of every
// branch of the tree), and leaf i is the one reached iff the outcomes of its ancestors are the ones on its path:
// (outcomes & leaf_mask) == leaf_want. The leaf's cluster is resolved on the host (context -> cluster -> hybrid config, alias table).
// Trees with at most 64 leaves, properties 0..14, no weighted predictor; rANS code specs without LZ77.
struct DevCoopTree {
	uint32_t used_props;        // bit q: some branch tests property q
	int32_t num_nodes, num_leaves, pad;
	int32_t node_prop[64];      // -1: no branch in this lane
	int32_t node_thr[64];
	uint32_t mask_lo[64], mask_hi[64], want_lo[64], want_hi[64];
	uint32_t leaf_a[64];        // predictor | hybrid-int config << 4 | max_token << 16
	uint32_t leaf_tab[64];      // the leaf's cluster: first alias entry in the u64 pool
	int32_t leaf_off[64], leaf_mul[64];
};

// a coded channel of the frame (or a plane of a section's sub-image): tightly packed int16 rows
struct DevPlaneRef { int16_t *ptr; int32_t w, h, meta, pad; };
// one channel of one section: a rectangle of plane `plane`; shifts = hshift | vshift << 8 of the channel (matched when the MA
// tree looks for "previous channels" of the same geometry)
struct DevChanRect { int32_t plane, x0, y0, w, h, shifts; };

struct DevSubPlane { int16_t *ptr; int32_t w, h, meta, pad; };

struct DevTransform { int32_t kind, begin_c, rct_type, num_c, nb_colours, nb_deltas, d_pred, pad; };

enum { MOD_MAX_CHANNELS = 256 };   // the reference's limit on channels while transforms are undone (j40.h:1173)

struct DevModFrame {
	int32_t width, height, num_groups, bpp;
	int32_t num_sections;           // LfGlobal's channel data (if any) + one per group
	int32_t num_channels;           // channels of the global Modular image as coded (before inverse transforms)
	int32_t tree_uses_wp, num_tree_nodes;
	int32_t max_width;              // widest rectangle any section decodes (sizes the weighted-predictor rows)
	int32_t check_section_end;      // see DevFrame::check_section_end
	uint32_t single_declared_end;   // see DevFrame::single_declared_end
};

struct DevModPlan {
	const DevModFrame *frame;
	const uint8_t *codestream;
	const uint8_t *pool_u8;
	const int32_t *pool_i32;
	const uint64_t *pool_u64;
	const DevCluster *clusters;
	const DevCodeSpec *spec;          // code specs: [0] the global one, then the sections' own (DevModSection::spec_idx)
	const DevTreeNode *tree;          // the global tree first, then the sections' own trees (DevModSection::tree_off)
	const DevModSection *sections;    // [num_sections]
	const int32_t *local_rct;         // {begin_c (section-relative), rct_type} pairs of the sections' own transforms
	const DevSubPlane *sub_planes;    // planes of the sections that decode into a sub-image of their own (DevModSection::sub_off)
	const DevPlaneRef *planes;        // [num_channels] sample planes of the coded channels; meta = 1: meta channel (palette), decoded
	                                  // whole and never a "previous channel" of image channels
	const DevChanRect *chan_rects;    // DevModSection::chan_off
	const DevCoopTree *coop_trees;    // DevModSection::coop_idx
	int32_t *wp_scratch;              // [num_sections][2 * max_width * 5] weighted-predictor error rows
	int32_t *lz_window; uint32_t lz_window_size;
	uint32_t *status;                 // [num_sections]
	int32_t *residuals;               // the split sections' residual tokens (DevModSection::res_off); also their LZ77 windows
	uint32_t *split_state;            // [num_sections][3]: the token pass's code and ordinal, the prediction pass's first overflowing ordinal
};

// one frame of a batch-wide launch of the pixel kernels (blockIdx.y): what the single-frame launch passes as kernel arguments
struct K2Frame {
	DevPlan plan;
	const DevVarblock *sorted;     // the frame's varblocks sorted by DctSelect
	float *large_scratch;
	uint8_t *rgba; size_t stride;  // where this decode writes
	int32_t class_start[28];
};

// one wavefront of the throughput-oriented K1: up to 64 consecutive groups of one frame of the batch
struct HfLaneWork { int32_t frame, first_group, num_groups, pad; };

// sRGB threshold table of the pixel kernels (idct_dev.h, srgb_u8_from_thresholds)
// 258 thresholds, then SRGB_BUCKETS bytes: the sample at the low end of each bucket of linear values, a bucket being the floats
// that share their top 16 bits (sign, exponent, 7 mantissa bits), from 2^-13 (SRGB_BUCKET_LO) up to 1.0 (SRGB_BUCKET_HI)
enum { SRGB_THRESHOLDS = 258, SRGB_BUCKET_LO = (127 - 13) << 7, SRGB_BUCKET_HI = 127 << 7, SRGB_BUCKETS = SRGB_BUCKET_HI - SRGB_BUCKET_LO + 4,
       SRGB_TABLE_FLOATS = SRGB_THRESHOLDS + SRGB_BUCKETS / 4 };

enum { HF_WAVES = 4 };
// k_hf_lanes keeps per wavefront, behind its frame's tables: the column state of the non-zero-count predictor ([3][32][64 lanes]
// bytes) and, when built with J40_LANE_EV_FLUSH > 0 (make EVENT_RING=8), the lanes' event rings (hf_lanes_dev.h: a lane's coefficient
// events collect in LDS and leave J40_LANE_EV_FLUSH at a time as one aligned 16- or 32-byte store; [2 * J40_LANE_EV_FLUSH][64 lanes]
// words). The default build has none -- every event is a 4-byte store of its own. Measured with rings of 2 x 8 (2 x 4) slots, 256
// 8K frames per launch: WRITE_SIZE 20.6 -> 10.5 (15.0) GB, but k_hf_lanes alone 41.9 -> 44.0 (45.2) ms, 32 (16) KB more LDS per
// workgroup of eight wavefronts -- 131 KB: no pixel-kernel workgroup but the 8x8 transform's fits beside it any more, and the queued
// form (512 frames per launch) gets two wavefronts per workgroup instead of four, 68 ms per 256 frames instead of 34. The writes were
// never what the kernel waited for (0.5 TB/s); the ring stays a build option (profiles/r05_event_ring_*.jsonl).
#ifndef J40_LANE_EV_FLUSH
#define J40_LANE_EV_FLUSH 0
#endif
enum { HF_LANE_PRED_BYTES = 3 * 32 * 64, HF_LANE_RING_SLOTS = J40_LANE_EV_FLUSH ? 2 * J40_LANE_EV_FLUSH : 1 /* (never indexed without rings) */,
       HF_LANE_COLS_BYTES = HF_LANE_PRED_BYTES + (J40_LANE_EV_FLUSH ? HF_LANE_RING_SLOTS * 64 * 4 : 0) };

// what the host knows about the entropy tables' sizes, to lay out K1's LDS
struct HfLaunchInfo {
	uint32_t block_ctx_size;     // bytes
	uint32_t max_num_dist;       // largest context count over the passes
	uint32_t max_clusters;
	uint32_t max_table_bytes;    // largest alias / prefix table span over the passes
	bool tables_fit_lds;
	bool lanes_fast;             // every pass: rANS without LZ77, and hf_lanes_dev.h's tables fit the LDS budget
	uint32_t lanes_lds_bytes;    // LDS k_hf_lanes needs for this frame's tables (plus HF_LANE_COLS_BYTES per wavefront)
};

// ---- LfGroup sections of VarDCT frames decoded on the device (lf_decode.hip) ----
// One LfGroup section = two Modular sub-images in one bit stream (j40.h:6722-6790): the LF coefficients (3 channels of w8 x h8,
// stream index sidx0), then the number of varblocks, a second Modular header and the HF metadata (x-from-y and b-from-y maps of
// w64 x h64, the varblock-info channel of 2 rows x nb_varblocks, the sharpness map of w8 x h8; stream index sidx2). The host
// reads what precedes the first stream (extra precision, first header) and hands over where it starts.
struct DevLfResult { uint32_t status; int32_t nb_varblocks; uint32_t raw_mask, stopped_at; };   // raw_mask / stopped_at: k_lf_rows' notes for k_lf_predict (lf_rows_dev.h), zero again once it has run
struct DevLfTask {
	// the frame the section belongs to (one launch takes the LfGroup sections of many frames): its codestream (padded), its
	// global MA tree laid out for the cooperative decoder, the alias tables of its global code spec
	const uint8_t *codestream; const DevCoopTree *tree; const uint64_t *alias; int32_t log_alpha_size;
	uint32_t byte_off, size, bit_off;
	int32_t w8, h8, w64, h64, sidx0, sidx2;
	int32_t nbvb_bits;      // ceil(log2(w8 * h8)): the width of the varblock count
	// where the decoded planes go: the LF coefficients in streamed order Y, X, B (w8 * h8 each), the chroma-from-luma maps (w64 * h64
	// each), the varblock-info channel (2 rows of nb_varblocks, pitch nb_varblocks; info_capacity int16 elements are reserved: a
	// section that claims more varblocks than that reports ERR_LFFB), the sharpness map (w8 * h8; nobody reads it)
	int16_t *lf[3], *xfromy, *bfromy, *info, *sharp;
	uint32_t info_capacity;
	DevLfResult *result;
};
// the LfGroup sections of one frame for k_lf_lanes (lf_lanes_dev.h: one section per lane): the tasks, and the frame's global MA tree
// and code spec in the lane decoders' table format (DevCodeSpec::lane_cfg_off)
struct DevLfLaneSet {
	const DevLfTask *tasks; int32_t ntasks;
	const DevTreeNode *tree; int32_t num_nodes;
	const uint8_t *ctx_map; int32_t num_dist;         // context -> cluster
	const uint32_t *cluster_cfg; int32_t num_clusters;
	const uint64_t *alias; int32_t log_alpha;
	uint32_t uses;                                      // bit 0: the tree looks at NE, 1: NEE, 2: NN
	uint32_t lds_bytes;                                 // what staging the tables takes
};
// A wavefront of k_lf_lanes: up to 64 sections, taken from up to LF_WAVE_PARTS frames (8K frames have 12 LfGroup sections each, 1080p
// frames one: a frame per wavefront would leave most lanes idle, and idle lanes cost what busy ones cost). A part is a run of
// sections of one DevLfLaneSet; the tables of every part are staged in LDS side by side.
enum { LF_WAVE_PARTS = 16 };
struct DevLfWave { int32_t num_parts, pad; struct { int32_t set, first_task, count, pad; } part[LF_WAVE_PARTS]; };
// Readable bytes behind a frame's codestream on the device: the lane decoders ask for up to three words past the position they stop
// at, and k_lf_rows' straight-line runs look at their position only when a run is over (lf_rows_dev.h, lf_row_deferred): a run is at
// most 255 samples of at most 32 bits each
enum { LF_CODESTREAM_PAD = 1280 };
enum { ERR_LFFB = ('l' << 24) | ('f' << 16) | ('f' << 8) | 'b' };   // not an error of the stream: the second Modular header is not the plain one the device handles; the host decodes this section

// ---- the LF-dependent half of a VarDCT frame's plan, built on the device (plan_dev.h, plan_kernels.hip) ----
// In the pipeline the host parses what precedes the LfGroup sections (headers, TOC, LfGlobal, HfGlobal) and the first bits of
// every LfGroup section; the sections' streams are decoded by k_lf_groups (or by a host thread, which then uploads the same raw
// planes), and everything the reference derives from them in j40__lf_group / j40__hf_metadata (j40.h:6722-6790, 6585-6720: LF
// index, varblock placement, quantisation-field index, coefficient offsets) plus the work lists of K1 / K2 (plan_build.cpp on
// the host path) is computed by three kernels: place (serial per LfGroup), scan (per frame), emit (per varblock).

// one LfGroup section of a frame: status of its streams and of the placement
struct DevLfSlot {
	uint32_t status;        // 0, the 4-char code of the section's streams (k_lf_groups / the host decoder), ERR_LFFB, or a placement error (vblk, dct?)
	int32_t nb_varblocks;   // as coded in the section (j40.h:6748)
	uint32_t dct_used;      // bit d: a varblock with DctSelect d was placed (j40.h:6640: which dequantisation matrices / orders the frame needs)
	int32_t placed;         // varblocks placed; == nb_varblocks when status == 0
};

// a placed varblock, in placement order (= the reference's varblock index); 16 bytes
struct DevVbRec {
	uint32_t coeffoff_qfidx;     // j40__varblock (j40.h:6352)
	int16_t hfmul_m1;
	uint8_t x8, y8;              // top-left cell inside the LfGroup (an LfGroup is at most 256 x 256 cells)
	uint8_t dctsel, grp;         // grp: (y8 >> 5) * 8 + (x8 >> 5), the group inside the LfGroup
	uint16_t rank_in_group;      // among the group's varblocks in raster order of their top-left cells: the visiting order of j40__hf_coeffs
	uint32_t rank_in_class;      // among the LfGroup's varblocks with the same DctSelect, in placement order
};

struct DevPlanBuild {
	// LfGlobal constants (frame.hpp: Frame)
	int32_t lf_thr[3][15], qf_thr[15], nb_lf_thr[3], nb_qf_thr;   // thresholds on the raw LF integers, channels X, Y, B (j40.h:6276-6290)
	int32_t num_lf_groups, num_groups, gcolumns, ggcolumns;
	int32_t lfidx_size;
	float mult_base, base_corr_x, base_corr_b, inv_colour_factor;
	uint32_t block_ctx_map_off;   // into pool_u8
	const uint8_t *pool_u8;
	DevLfGroup *lf_groups;        // the plan's array: nb_varblocks is filled in by the placement
	DevLfSlot *lf_slots;          // [num_lf_groups]
	const int16_t *lfraw[3];      // decoded LF integers per cell, channels X, Y, B (DevPlan::lfraw)
	const int16_t *xfromy, *bfromy;   // per 64x64 cell, at DevLfGroup::c64_base
	// the varblock-info channel of every LfGroup: two rows of nb_varblocks samples (DctSelect; HfMul - 1) at 2 * cell_base, the
	// second row nb_varblocks after the first (the channel's own pitch)
	const int16_t *vbinfo;
	DevVbRec *vb_recs;            // at DevLfGroup::vb_base
	uint32_t *group_count;        // [num_groups] varblocks per group
	uint32_t *group_block_start;  // [num_groups + 1] (the plan's array)
	uint32_t *class_count;        // [num_lf_groups][28] varblocks per DctSelect value; the scan turns it into each (LfGroup, class)'s first index in vb_sorted
	int32_t *class_start;         // [28] (27 = the number of varblocks)
	DevGroupBlock *group_blocks;  // the plan's arrays
	DevVarblock *vb_sorted;
	// [0] the frame's verdict over LfGroup and pass-group sections (first failing section in file order), [1] flags (bit 0: some
	// LfGroup reported ERR_LFFB, bit 1: some section ERR_EVOF), [2] union of DevLfSlot::dct_used, [3] varblocks
	uint32_t *verdict;
	const uint32_t *lf_section_off;   // [num_lf_groups] byte offset of every LfGroup section (the order the reference reads them in)
	// the LfGroup tail (lf_tail_kernels.hip): scratch of three planes of `cells` floats for the dequantised, smoothed LF samples
	float *lf_scratch; float inv_m_lf[3]; int32_t lf_smooth; uint32_t cells;
};
// one LfGroup of a batch of frames
struct DevBatchLf { int32_t frame, lfg; };

// sizes the host knows about a Modular frame's tree and code tables, to lay out k_modular_sections' LDS
struct ModLaunchInfo {
	int32_t num_tree_nodes, num_dist, num_clusters; uint32_t table_bytes; int32_t max_width, uses_wp;
	int32_t coop_width;   // widest channel rectangle of the sections k_modular_coop decodes; 0: none
	int32_t all_coop;     // every section goes to k_modular_coop or k_modular_quad
	// k_modular_quad: sections flagged `quad` (0: none), the code spec they all decode with, their widest channel
	int32_t quad_sections, quad_spec, quad_width;
	int32_t coop_sections;   // sections with coop_idx >= 0 (quad ones included)
	uint32_t quad_table_span; // alias entries of quad_spec
	// modular_split.hip: sections flagged `split` (0: none), their widest channel, the most channels one of them codes
	int32_t split_sections, split_width, split_channels;
};

enum {
	ERR_SHRT = ('s' << 24) | ('h' << 16) | ('r' << 8) | 't',
	ERR_COEF = ('c' << 24) | ('o' << 16) | ('e' << 8) | 'f',
	ERR_EXCS = ('e' << 24) | ('x' << 16) | ('c' << 8) | 's',
	ERR_ANS = ('a' << 24) | ('n' << 16) | ('s' << 8) | '?',
	ERR_IOVF = ('i' << 24) | ('o' << 16) | ('v' << 8) | 'f',
	ERR_PAD0 = ('p' << 24) | ('a' << 16) | ('d' << 8) | '0',
	ERR_RNGE = ('r' << 24) | ('n' << 16) | ('g' << 8) | 'e',
	ERR_POVF = ('p' << 24) | ('o' << 16) | ('v' << 8) | 'f',
	ERR_PRED = ('p' << 24) | ('r' << 16) | ('e' << 8) | 'd',
	ERR_TREC = ('t' << 24) | ('r' << 16) | ('e' << 8) | 'c',
	ERR_TODO = ('T' << 24) | ('O' << 16) | ('D' << 8) | 'O',
	ERR_VBLK = ('v' << 24) | ('b' << 16) | ('l' << 8) | 'k',
	ERR_DCTQ = ('d' << 24) | ('c' << 16) | ('t' << 8) | '?',
	ERR_EVOF = ('e' << 24) | ('v' << 16) | ('o' << 8) | 'f',   // a section's event region is full, or a coefficient does not fit an event's 16 bits: decode the frame with dense planes
};

// the pixel rectangles {x0, y0, x1, y1} a contiguous range of groups (raster order) covers: at most three -- the tail of its first
// group row, whole group rows, the head of its last group row (j40_amd/sharding.py range_rectangles is the same arithmetic)
static inline int group_range_rects(int64_t first, int64_t count, int32_t width, int32_t height, int32_t group_shift, int32_t rects[3][4]) {
	const int64_t dim = (int64_t) 1 << group_shift, gcols = (width + dim - 1) / dim;
	int n = 0;
	for (int64_t g = first, end = first + count; g < end && n < 3; ) {
		const int64_t row = g / gcols, col = g % gcols;
		int64_t x0, y0, x1, y1;
		if (col == 0 && end - g >= gcols) { const int64_t rows = (end - g) / gcols; x0 = 0; y0 = row * dim; x1 = width; y1 = (row + rows) * dim; g += rows * gcols; }
		else { const int64_t k = (gcols - col < end - g) ? gcols - col : end - g; x0 = col * dim; y0 = row * dim; x1 = (col + k) * dim; y1 = (row + 1) * dim; g += k; }
		rects[n][0] = (int32_t) x0; rects[n][1] = (int32_t) y0; rects[n][2] = (int32_t) (x1 < width ? x1 : width); rects[n][3] = (int32_t) (y1 < height ? y1 : height);
		++n;
	}
	return n;
}

} // namespace j40hip
