// j40_amd/csrc/frame.hpp -- host-side parse of everything in front of the per-group hot path:
// container, image metadata, frame header, TOC, LfGlobal, HfGlobal/HfPass and the LfGroup sections
// (LF image, varblock layout, LLF coefficients). Produces the tables the HIP kernels consume.
//
// Reference behaviour (all host-only rows of SURVEY.md section 2): j40__container (j40.h:1479),
// j40__image_metadata (3104), j40__frame_header (5163), j40__read_toc (5479), j40__lf_global (6257),
// j40__lf_group (6722), j40__hf_global (6819), j40__load_dq_matrix (4828), j40__natural_order (4980).
#pragma once
#include "modular.hpp"
#include <array>

namespace j40hip {

struct ExtraChannel { int32_t type = 0, bpp = 8, exp_bits = 0, dim_shift = 0; bool alpha_associated = false; };
enum { EC_ALPHA = 0, EC_SPOT = 2, EC_BLACK = 4, EC_CFA = 5 };

struct ImageMeta {
	int32_t width = 0, height = 0;
	int32_t bpp = 8, exp_bits = 0;
	bool have_animation = false, anim_have_timecodes = false;
	bool modular_16bit_buffers = true;
	std::vector<ExtraChannel> ec;
	bool xyb_encoded = true, want_icc = false, grey = false;
	float intensity_target = 255.0f;
	float opsin_inv_mat[3][3];
	float opsin_bias[3];
	float quant_bias[3];
	float quant_bias_num = 0.145f;
};

struct FrameHeader {
	bool is_last = true;
	int32_t type = 0;
	bool is_modular = false;
	bool has_noise = false, has_patches = false, has_splines = false, use_lf_frame = false, skip_adapt_lf_smooth = false;
	bool do_ycbcr = false;
	int32_t jpeg_upsampling = 0;
	int32_t group_size_shift = 8;
	int32_t x_qm_scale = 3, b_qm_scale = 2;
	int32_t num_passes = 1;
	int32_t x0 = 0, y0 = 0, width = 0, height = 0;
	int32_t grows = 0, gcolumns = 0, ggrows = 0, ggcolumns = 0;
	int64_t num_groups = 0, num_lf_groups = 0;
	// RestorationFilter as the reference parses it (defaults j40.h:5196-5208, fields j40.h:5339-5366). The reference reads it and never
	// looks at it again; here it is kept for the restoration filters (device/restore_dev.h), which run only when asked for.
	struct Restoration {
		bool gab = true;
		float gab_weights[3][2] = {{0.115169525f, 0.061248592f}, {0.115169525f, 0.061248592f}, {0.115169525f, 0.061248592f}};
		int32_t epf_iters = 2;
		float sharp_lut[8] = {0.0f / 7.0f, 1.0f / 7.0f, 2.0f / 7.0f, 3.0f / 7.0f, 4.0f / 7.0f, 5.0f / 7.0f, 6.0f / 7.0f, 7.0f / 7.0f};
		float channel_scale[3] = {40.0f, 5.0f, 3.5f};
		float quant_mul = 0.46f, pass0_sigma_scale = 0.9f, pass2_sigma_scale = 6.5f, border_sad_mul = 2.0f / 3.0f, sigma_for_modular = 1.0f;
	} restoration;
};

struct Section { size_t offset = 0, size = 0; };  // byte range inside the codestream

struct Toc {
	bool single = false;
	Section single_section;               // when the frame has exactly one section: readable to the end of the codestream like in the
	                                      // reference, which reads such a frame from its main state (no section boundary) ...
	size_t single_declared_end = 0;       // ... and compares where it ended with the TOC entry afterwards (j40__end_of_frame, j40.h:7796)
	Section lf_global, hf_global;
	std::vector<Section> lf_groups;       // [num_lf_groups]
	std::vector<Section> pass_groups;     // [num_passes * num_groups], pass-major
	size_t end_offset = 0;
};

struct DqMatrix {
	int32_t mode = 0;                     // 0 library, 7 raw, else the coded parameter form
	int32_t n = 0, m = 0;
	std::vector<std::array<float, 3>> params;
	bool loaded = false;                  // expanded to one weight per coefficient
};

struct VarblockInfo { int32_t coeffoff_qfidx; float hfmul_inv; int32_t x8, y8, dctsel; };

struct LfGroup {
	int32_t idx = 0, left = 0, top = 0, width = 0, height = 0, width8 = 0, height8 = 0, width64 = 0, height64 = 0;
	std::vector<int32_t> blocks;          // [height8 * width8]: (dctsel + 2) << 20 | varblock, 1 << 20 | varblock
	std::vector<uint8_t> lfindices;       // [height8 * width8]
	std::vector<VarblockInfo> varblocks;
	std::vector<float> llfcoeffs[3];      // [height8 * width8], indexed by coefficient offset / 64
	std::vector<int16_t> xfromy, bfromy;  // [height64 * width64]
	std::vector<int16_t> sharpness;       // [height8 * width8], as decoded (the reference's j40__lf_group_st::sharpness; read by the edge-preserving filter only)
	bool loaded = false;
	// Frame::defer_lf_tail: the quantised LF samples as decoded (channel order X, Y, B) and their dequantisation factors; the
	// dequantisation, the adaptive smoothing and the LLF coefficients (`llfcoeffs`) are then computed on the device at upload
	// (finish_lf_tail does the same on the host when somebody asks for llfcoeffs)
	std::vector<int16_t> lfraw[3];
	float mult_lf[3] = {0.0f, 0.0f, 0.0f};
	bool tail_pending = false;
};

// One LfGroup section handed to a device decoder (Frame::lf_decoder; device/lf_decode.hip): the host has read what precedes the
// LF coefficient stream; the decoder fills in the results (plane pointers stay valid until its next call on the same thread).
struct LfDeviceTask {
	size_t byte_off = 0, size = 0; uint32_t bit_off = 0;
	int32_t w8 = 0, h8 = 0, w64 = 0, h64 = 0, sidx0 = 0, sidx2 = 0, nbvb_bits = 0;
	uint32_t status = 0;          // 0, the stream's 4-char error, or 'lffb': decode this section on the host
	int32_t nb_varblocks = 0;
	const int16_t *lf[3] = {nullptr, nullptr, nullptr};   // streamed order Y, X, B
	const int16_t *xfromy = nullptr, *bfromy = nullptr, *info0 = nullptr, *info1 = nullptr, *sharp = nullptr;
};
// the streams of one LfGroup section as decoded (read_lf_group_raw): LF integers in streamed order Y, X, B; chroma-from-luma
// maps; the varblock-info channel (two rows of nb_varblocks: DctSelect, HfMul - 1); the sharpness map.
struct LfRaw {
	int32_t extra_prec = 0, nb_varblocks = 0;
	std::vector<int16_t> lf[3], xfromy, bfromy, info, sharp;
};
struct Frame;
// returns false when it cannot take the frame (tree / code spec outside what the kernel handles, no device): host path
typedef bool (*LfDeviceDecoder)(void *ctx, const Frame &f, const uint8_t *cs, size_t cs_size, std::vector<LfDeviceTask> &tasks);

struct Frame {
	ImageMeta im;
	FrameHeader fh;
	Toc toc;

	// LfGlobal
	float m_lf_scaled[3] = {1.0f / 4096.0f, 1.0f / 512.0f, 1.0f / 256.0f};
	int32_t global_scale = 0, quant_lf = 0;
	int32_t lf_thr[3][15], qf_thr[15];
	int32_t nb_lf_thr[3] = {0, 0, 0}, nb_qf_thr = 0;
	std::vector<uint8_t> block_ctx_map;
	int32_t nb_block_ctx = 0;
	float inv_colour_factor = 1.0f / 84.0f, base_corr_x = 0.0f, base_corr_b = 1.0f;
	int32_t x_factor_lf = 0, b_factor_lf = 0;
	std::vector<TreeNode> global_tree;
	CodeSpec global_codespec;
	Modular gmodular;                     // frame-sized Modular image (Modular frames, extra channels)
	int32_t num_gm_channels = 0;          // channels already decoded inside LfGlobal
	// where the not-yet-decoded global Modular pixel data would start is irrelevant: it is decoded here

	// HfGlobal / HfPass
	DqMatrix dq_matrix[17];
	int32_t num_hf_presets = 0;
	std::vector<int32_t> order_lehmer[11][13][3];
	bool order_has_lehmer[11][13][3];
	std::vector<int32_t> orders[11][13][3];   // expanded coefficient orders (only those in use)
	CodeSpec coeff_codespec[11];
	uint32_t dct_select_used = 0, order_used = 0;

	std::vector<LfGroup> lf_groups;
	size_t single_pass_group_bitpos = 0;
	// VarDCT frames: leave the tail of every LfGroup (dequantisation, adaptive smoothing, LLF coefficients; j40.h:6544-6590, 6492, 5944)
	// to the device: the host keeps the decoded integers (LfGroup::lfraw). Set before parse_frame.
	bool defer_lf_tail = false;
	// VarDCT frames with several sections: the LfGroup streams are decoded by this (on the device) instead of the host. Set before parse_frame.
	LfDeviceDecoder lf_decoder = nullptr; void *lf_decoder_ctx = nullptr;
	bool lf_decoded_on_device = false;   // out: it was
	// Streaming input (the reference's refillable source, j40.h:1220-1386, 1676-1812): the codestream's bytes arrive
	// while it is parsed. need_bytes(ctx, n) returns once bytes [0, n) of the buffer are there (or the source has ended: what is
	// missing then reads as a truncated stream). parse_frame asks before it touches anything: the headers on growing prefixes, then
	// LfGlobal, HfGlobal and every LfGroup section as their bytes are due -- the pass-group sections, most of the file, are not the
	// host's to read. nullptr: everything is there. Set before parse_frame; called from the LfGroup worker threads too.
	void (*need_bytes)(void *ctx, size_t upto) = nullptr; void *need_ctx = nullptr;
	size_t (*have_bytes)(void *ctx) = nullptr;   // how many bytes are there now (with need_bytes)
	void need(size_t upto) const { if (need_bytes) need_bytes(need_ctx, upto); }
	// A fresh Frame with the fields a caller sets BEFORE parse_frame -- the ones declared above, from defer_lf_tail on -- and nothing
	// else (the streaming header parse starts over with one when the prefix it had ran out). A field added to that set goes in here.
	Frame with_same_inputs() const {
		Frame g;
		g.defer_lf_tail = defer_lf_tail; g.lf_decoder = lf_decoder; g.lf_decoder_ctx = lf_decoder_ctx;
		g.need_bytes = need_bytes; g.need_ctx = need_ctx; g.have_bytes = have_bytes;
		return g;
	}
	// Modular frames: LfGlobal's channel data is left to the device; it starts at this bit of the section
	bool gm_data_pending = false;
	size_t gm_data_bitpos = 0;  // single-section VarDCT frames: where the pass group starts
};

// locates the codestream inside `data` (bare codestream or ISOBMFF container); if the codestream is
// split over several boxes it is reassembled into `storage`
// stray_tail (optional): 1..7 when that many bytes follow the last box of a container -- too few for a box header
void extract_codestream(const uint8_t *data, size_t size, const uint8_t **cs, size_t *cs_size, std::vector<uint8_t> *storage, int *stray_tail = nullptr);

// parses headers, TOC, LfGlobal, HfGlobal and every LfGroup section. `threads` > 1 decodes LfGroup
// sections concurrently (they are independent given LfGlobal)
void parse_frame(const uint8_t *cs, size_t cs_size, Frame *f, int threads);
// the pipeline's host stage (see frame.cpp)
bool parse_frame_front(const uint8_t *cs, size_t cs_size, Frame *f, std::vector<LfDeviceTask> *tasks, std::vector<int32_t> *extra_prec, bool *plain);
void read_lf_group_raw(BitReader &br, const Frame &f, const LfGroup &gg, LfRaw *out);
// the LfGroup tail on the host for the groups that still have it pending (dequantise, smooth, LLF): what the device does at upload
void finish_lf_tail(Frame *f);

struct GroupInfo { int32_t ggidx, gx_in_gg, gy_in_gg, gw, gh; };
GroupInfo group_info(const FrameHeader &fh, int64_t gidx);  // j40.h:7734

struct DctSelect { int8_t log_rows, log_columns, param_idx, order_idx; };
extern const DctSelect DCT_SELECT[27];
extern const int8_t LOG_ORDER_SIZE[13][2];

void natural_order(int32_t log_rows, int32_t log_columns, std::vector<int32_t> *out);
void load_dq_matrix(int32_t idx, DqMatrix *dq);
void forward_dct2d_scaled_for_llf(float *buf, float *scratch, int32_t log_rows, int32_t log_columns);

} // namespace j40hip
