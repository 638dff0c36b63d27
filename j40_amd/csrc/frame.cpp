// j40_amd/csrc/frame.cpp -- see frame.hpp
#include "frame.hpp"
#include "tables.hpp"
#include <algorithm>
#include <cmath>
#include <thread>
#include <atomic>
#include <chrono>
#include <mutex>

namespace j40hip {

// ------------------------------------------------------------------------------------------------
// container (ISO 18181-2 boxes; reference: j40__container, j40.h:1479)

static uint32_t be32(const uint8_t *p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }

void extract_codestream(const uint8_t *data, size_t size, const uint8_t **cs, size_t *cs_size, std::vector<uint8_t> *storage, int *stray_tail) {
	if (stray_tail) *stray_tail = 0;
	static const uint8_t SIGNATURE[32] = {0, 0, 0, 0x0c, 'J', 'X', 'L', ' ', 0x0d, 0x0a, 0x87, 0x0a, 0, 0, 0, 0x14, 'f', 't', 'y', 'p', 'j', 'x', 'l', ' ', 0, 0, 0, 0, 'j', 'x', 'l', ' '};
	J40HIP_SHOULD(size >= 2, "shrt");
	if (data[0] == 0xff && data[1] == 0x0a) { *cs = data; *cs_size = size; return; }
	J40HIP_SHOULD(data[0] == SIGNATURE[0] && data[1] == SIGNATURE[1], "!jxl");
	J40HIP_SHOULD(size >= 32, "shrt");
	J40HIP_SHOULD(memcmp(data, SIGNATURE, 12) == 0, "!jxl");
	J40HIP_SHOULD(memcmp(data + 12, SIGNATURE + 12, 20) == 0, "ftyp");
	size_t pos = 32;
	bool seen_jxlc = false, seen_jxlp = false, seen_jxll = false, seen_jxli = false;
	struct Piece { size_t off, len; };
	std::vector<Piece> pieces;
	// The reference maps boxes on demand (j40__container, j40.h:1479): once the codestream is complete it never looks at what
	// follows, and a codestream box that is cut off yields the bytes that are there (the decoder then runs out of data in whatever
	// section that hits). Same here: stop behind `jxlc` / the last `jxlp`, and take a truncated codestream box as far as it goes.
	bool complete = false, no_more_codestream = false;
	while (pos < size) {
		if (size - pos < 8) { J40HIP_SHOULD(!pieces.empty(), "shrt"); if (stray_tail) *stray_tail = (int) (size - pos); break; }
		if (complete) break;
		uint64_t box = be32(data + pos);
		uint32_t type = be32(data + pos + 4);
		size_t header = 8, payload;
		if (box == 1) {
			J40HIP_SHOULD(size - pos >= 16, "shrt");
			box = ((uint64_t) be32(data + pos + 8) << 32) | be32(data + pos + 12);
			J40HIP_SHOULD(box >= 16, "boxx");
			header = 16;
		} else if (box != 0) J40HIP_SHOULD(box >= 8, "boxx");
		bool to_eof = box == 0;
		payload = to_eof ? size - pos - header : (size_t) box - header;
		const bool cut_off = !to_eof && payload > size - pos - header;
		if (cut_off) {
			if (type != 0x6a786c63 && type != 0x6a786c70) { J40HIP_SHOULD(!pieces.empty(), "shrt"); break; }   // junk behind the codestream
			payload = size - pos - header; to_eof = true;
		}
		size_t body = pos + header;
		switch (type) {
		case 0x6a786c6c: J40HIP_SHOULD(!seen_jxll, "box?"); seen_jxll = true; break;          // jxll
		case 0x6a786c69: J40HIP_SHOULD(!seen_jxli, "box?"); seen_jxli = true; break;          // jxli
		case 0x6a786c63:                                                                       // jxlc
			J40HIP_SHOULD(!seen_jxlc && !seen_jxlp, "box?");
			seen_jxlc = true; pieces.push_back({body, payload}); complete = true;
			break;
		case 0x6a786c70:                                                                       // jxlp
			J40HIP_SHOULD(!seen_jxlc && !no_more_codestream, "box?");
			J40HIP_SHOULD(payload >= 4, "jxlp");
			seen_jxlp = true; pieces.push_back({body + 4, payload - 4});                       // the sequence index is not interpreted ...
			if (!(data[body] >> 7)) no_more_codestream = true;                                 // ... except for this (j40.h:1557, as the reference has it)
			break;
		case 0x62726f62:                                                                       // brob
			J40HIP_SHOULD(payload > 4, "brot");
			{ uint32_t inner = be32(data + body); J40HIP_SHOULD(inner != 0x62726f62 && (inner >> 8) != 0x6a786c, "brot"); }
			break;
		default: break;
		}
		if (to_eof) break;
		pos = body + payload;
	}
	J40HIP_SHOULD(!pieces.empty(), "shrt");
	if (pieces.size() == 1) { *cs = data + pieces[0].off; *cs_size = pieces[0].len; return; }
	storage->clear();
	for (const Piece &p : pieces) storage->insert(storage->end(), data + p.off, data + p.off + p.len);
	*cs = storage->data(); *cs_size = storage->size();
}

// ------------------------------------------------------------------------------------------------
// image metadata (j40.h:3001-3313)

static void read_size_header(BitReader &br, int32_t *w, int32_t *h) {  // j40.h:3008
	bool div8 = br.u(1);
	*h = div8 ? ((int32_t) br.u(5) + 1) * 8 : br.u32(1, 9, 1, 13, 1, 18, 1, 30);
	switch (br.u(3)) {
	case 0: *w = div8 ? ((int32_t) br.u(5) + 1) * 8 : br.u32(1, 9, 1, 13, 1, 18, 1, 30); break;
	case 1: *w = *h; break;
	case 2: *w = (int32_t) ((uint64_t) *h * 6 / 5); break;
	case 3: *w = (int32_t) ((uint64_t) *h * 4 / 3); break;
	case 4: *w = (int32_t) ((uint64_t) *h * 3 / 2); break;
	case 5: *w = (int32_t) ((uint64_t) *h * 16 / 9); break;
	case 6: *w = (int32_t) ((uint64_t) *h * 5 / 4); break;
	default: J40HIP_SHOULD(*h < 0x40000000, "bigg"); *w = *h * 2; break;
	}
}

static void read_bit_depth(BitReader &br, int32_t *bpp, int32_t *exp_bits) {  // j40.h:3033
	if (br.u(1)) {
		*bpp = br.u32(32, 0, 16, 0, 24, 0, 1, 6);
		*exp_bits = (int32_t) br.u(4) + 1;
		int32_t mantissa = *bpp - *exp_bits - 1;
		J40HIP_SHOULD(2 <= mantissa && mantissa <= 23, "bpp?");
		J40HIP_SHOULD(2 <= *exp_bits && *exp_bits <= 8, "exp?");
	} else {
		*bpp = br.u32(8, 0, 10, 0, 12, 0, 1, 6);
		*exp_bits = 0;
		J40HIP_SHOULD(1 <= *bpp && *bpp <= 31, "bpp?");
	}
}

// j40__name (j40.h:3050): the name is validated as UTF-8 the way the reference does it -- including `i + c < len` where `<=` was
// meant, which makes the last character of every name fail: any non-empty name is a `name` error there, and so it is here
static void skip_name(BitReader &br) {
	int32_t len = br.u32(0, 0, 0, 4, 16, 5, 48, 10);
	std::vector<uint8_t> buf((size_t) len + 1, 0);
	for (int32_t i = 0; i < len; ++i) buf[(size_t) i] = (uint8_t) br.u(8);
	for (int32_t i = 0; i < len; ) {
		int32_t c = buf[(size_t) i++];
		const int32_t cc = buf[(size_t) i];   // the terminating zero keeps this in range
		c = c < 0x80 ? 0 : c < 0xc2 ? -1 : c < 0xe0 ? 1 :
			c < 0xf0 ? ((c == 0xe0 ? cc >= 0xa0 : c == 0xed ? cc < 0xa0 : true) ? 2 : -1) :
			c < 0xf5 ? ((c == 0xf0 ? cc >= 0x90 : c == 0xf4 ? cc < 0x90 : true) ? 3 : -1) : -1;
		J40HIP_SHOULD(c >= 0 && i + c < len, "name");
		while (c-- > 0) J40HIP_SHOULD((buf[(size_t) i++] & 0xc0) == 0x80, "name");
	}
}

static void read_extensions(BitReader &br) {  // j40.h:3088
	uint64_t extensions = br.u64();
	int64_t nbits = 0;
	for (int i = 0; i < 64; ++i) if (extensions >> i & 1) {
		uint64_t n = br.u64();
		J40HIP_SHOULD(n <= (uint64_t) INT64_MAX - (uint64_t) nbits, "flen");
		nbits += (int64_t) n;
	}
	br.skip_bits_like_reference(nbits);
}

static void read_customxy(BitReader &br) {
	(void) br.u32(0, 19, 0x80000, 19, 0x100000, 20, 0x200000, 21);
	(void) br.u32(0, 19, 0x80000, 19, 0x100000, 20, 0x200000, 21);
}

static void read_image_metadata(BitReader &br, ImageMeta *im) {  // j40.h:3104
	static const float OPSIN_INV[3][3] = {
		{11.031566901960783f, -9.866943921568629f, -0.16462299647058826f},
		{-3.254147380392157f, 4.418770392156863f, -0.16462299647058826f},
		{-3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f}};
	memcpy(im->opsin_inv_mat, OPSIN_INV, sizeof OPSIN_INV);
	im->opsin_bias[0] = im->opsin_bias[1] = im->opsin_bias[2] = -0.0037930732552754493f;
	im->quant_bias[0] = 1.0f - 0.05465007330715401f;
	im->quant_bias[1] = 1.0f - 0.07005449891748593f;
	im->quant_bias[2] = 1.0f - 0.049935103337343655f;
	im->quant_bias_num = 0.145f;

	read_size_header(br, &im->width, &im->height);
	// Main profile, level 5 limits (j40.h:1170)
	J40HIP_SHOULD(im->width <= (1 << 18) && im->height <= (1 << 18), "slim");
	J40HIP_SHOULD((int64_t) im->width * im->height <= (1 << 28), "slim");

	if (!br.u(1)) {  // !all_default
		bool extra_fields = br.u(1);
		if (extra_fields) {
			(void) br.u(3);  // orientation
			if (br.u(1)) { int32_t w, h; read_size_header(br, &w, &h); }
			if (br.u(1)) J40HIP_RAISE("TODO");  // preview
			if (br.u(1)) {  // animation
				(void) br.u32(100, 0, 1000, 0, 1, 10, 1, 30);
				(void) br.u32(1, 0, 1001, 0, 1, 8, 1, 10);
				(void) br.u32_64(0, 0, 0, 3, 0, 16, 0, 32);
				im->have_animation = true;
				im->anim_have_timecodes = br.u(1);
			}
		}
		read_bit_depth(br, &im->bpp, &im->exp_bits);
		J40HIP_SHOULD(im->bpp <= 16, "fbpp");
		im->modular_16bit_buffers = br.u(1);
		J40HIP_SHOULD(im->modular_16bit_buffers, "fm32");
		int32_t num_ec = br.u32(0, 0, 1, 0, 2, 4, 1, 12);
		J40HIP_SHOULD(num_ec <= 4, "elim");
		im->ec.assign((size_t) num_ec, ExtraChannel());
		for (ExtraChannel &ec : im->ec) {
			if (br.u(1)) { ec.type = EC_ALPHA; ec.bpp = 8; }
			else {
				ec.type = br.enum_();
				read_bit_depth(br, &ec.bpp, &ec.exp_bits);
				ec.dim_shift = br.u32(0, 0, 3, 0, 4, 0, 1, 3);
				skip_name(br);
				switch (ec.type) {
				case EC_ALPHA: ec.alpha_associated = br.u(1); break;
				case EC_SPOT: for (int i = 0; i < 4; ++i) (void) br.f16(); break;
				case EC_CFA: (void) br.u32(1, 0, 0, 2, 3, 4, 19, 8); break;
				case EC_BLACK: J40HIP_RAISE("fblk");
				case 1: case 3: case 6: case 15: case 16: break;
				default: J40HIP_RAISE("ect?");
				}
			}
			J40HIP_SHOULD(ec.bpp <= 16, "fbpp");
		}
		im->xyb_encoded = br.u(1);
		if (!br.u(1)) {  // ColourEncoding
			im->want_icc = br.u(1);
			int32_t cspace = br.enum_();
			J40HIP_SHOULD(cspace <= 3, "csp?");
			im->grey = cspace == 1;
			if (!im->want_icc) {
				if (cspace != 2) {
					int32_t wp = br.enum_();
					J40HIP_SHOULD(wp == 1 || wp == 2 || wp == 10 || wp == 11, "wpt?");
					if (wp == 2) read_customxy(br);
					if (cspace != 1) {
						int32_t pr = br.enum_();
						J40HIP_SHOULD(pr == 1 || pr == 2 || pr == 9 || pr == 11, "prm?");
						if (pr == 2) { read_customxy(br); read_customxy(br); read_customxy(br); }
					}
				}
				if (br.u(1)) {
					int32_t gamma = (int32_t) br.u(24);
					J40HIP_SHOULD(gamma > 0 && gamma <= 10000000, "gama");
					if (cspace == 2) J40HIP_SHOULD(gamma == 3333333, "gama");
				} else {
					int32_t tf = br.enum_();
					J40HIP_SHOULD(tf == 1 || tf == 2 || tf == 8 || tf == 13 || tf == 16 || tf == 17 || tf == 18, "tfn?");
				}
				int32_t intent = br.enum_();
				J40HIP_SHOULD(intent <= 3, "itt?");
			}
		}
		if (extra_fields) {
			if (!br.u(1)) {  // ToneMapping
				im->intensity_target = br.f16();
				J40HIP_SHOULD(im->intensity_target > 0, "tone");
				float min_nits = br.f16();
				J40HIP_SHOULD(0 < min_nits && min_nits <= im->intensity_target, "tone");
				bool relative = br.u(1);
				float linear_below = br.f16();
				if (relative) J40HIP_SHOULD(0 <= linear_below && linear_below <= 1, "tone");
				else J40HIP_SHOULD(0 <= linear_below, "tone");
			}
		}
		read_extensions(br);
	}
	if (!br.u(1)) {  // !default_m
		if (im->xyb_encoded) {
			for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) im->opsin_inv_mat[i][j] = br.f16();
			for (int i = 0; i < 3; ++i) im->opsin_bias[i] = br.f16();
			for (int i = 0; i < 3; ++i) im->quant_bias[i] = br.f16();
			im->quant_bias_num = br.f16();
		}
		J40HIP_SHOULD(br.u(3) == 0, "TODO");  // custom upsampling weights
	}
}

// ------------------------------------------------------------------------------------------------
// frame header (j40.h:5163)

static void read_frame_header(BitReader &br, const ImageMeta &im, FrameHeader *f) {
	f->width = im.width; f->height = im.height;
	br.zero_pad_to_byte();
	if (!br.u(1)) {
		bool full_frame = true;
		f->type = (int32_t) br.u(2);
		f->is_modular = br.u(1);
		uint64_t flags = br.u64();
		f->has_noise = flags & 1; f->has_patches = flags >> 1 & 1; f->has_splines = flags >> 4 & 1;
		f->use_lf_frame = flags >> 5 & 1; f->skip_adapt_lf_smooth = flags >> 7 & 1;
		if (!im.xyb_encoded) f->do_ycbcr = br.u(1);
		if (!f->use_lf_frame) {
			if (f->do_ycbcr) f->jpeg_upsampling = (int32_t) br.u(6);
			J40HIP_SHOULD(br.u(2) == 0, "TODO");  // upsampling
			for (size_t i = 0; i < im.ec.size(); ++i) J40HIP_SHOULD(br.u(2) == 0, "TODO");
		}
		if (f->is_modular) f->group_size_shift = 7 + (int32_t) br.u(2);
		else if (im.xyb_encoded) { f->x_qm_scale = (int32_t) br.u(3); f->b_qm_scale = (int32_t) br.u(3); }
		if (f->type != 2) {
			f->num_passes = br.u32(1, 0, 2, 0, 3, 0, 4, 3);
			if (f->num_passes > 1) {
				int32_t num_ds = br.u32(0, 0, 1, 0, 2, 0, 3, 1), prev_ds = 4, ppass = 0;
				J40HIP_SHOULD(num_ds < f->num_passes, "pass");
				for (int32_t i = 0; i < f->num_passes - 1; ++i) (void) br.u(2);  // shift
				for (int32_t i = 0; i < num_ds; ++i) { int32_t ds = (int32_t) br.u(2); J40HIP_SHOULD(prev_ds >= ds, "pass"); prev_ds = ds; }
				for (int32_t i = 0; i < num_ds; ++i) {
					int32_t pass = br.u32(0, 0, 1, 0, 2, 0, 0, 3);
					J40HIP_SHOULD(i > 0 ? ppass < pass && pass < f->num_passes : pass == 0, "pass");
					ppass = pass;
				}
			}
		}
		if (f->type == 1) {
			(void) br.u(2);  // lf_level
		} else if (br.u(1)) {  // have_crop
			if (f->type != 2) {
				f->x0 = unpack_signed(br.u32(0, 8, 256, 11, 2304, 14, 18688, 30));
				f->y0 = unpack_signed(br.u32(0, 8, 256, 11, 2304, 14, 18688, 30));
			}
			f->width = br.u32(0, 8, 256, 11, 2304, 14, 18688, 30);
			f->height = br.u32(0, 8, 256, 11, 2304, 14, 18688, 30);
			J40HIP_SHOULD(f->width <= (1 << 18) && f->height <= (1 << 18), "slim");
			J40HIP_SHOULD((int64_t) f->width * f->height <= (1 << 28), "slim");
			full_frame = f->x0 <= 0 && f->y0 <= 0 && f->width + f->x0 >= im.width && f->height + f->y0 >= im.height;
		}
		int32_t blend_mode0 = 0, save_as_ref = 0;
		int64_t duration = 0;
		if (f->type == 0 || f->type == 3) {
			for (int32_t i = -1; i < (int32_t) im.ec.size(); ++i) {
				int32_t mode = br.u32(0, 0, 1, 0, 2, 0, 3, 2);
				if (i < 0) blend_mode0 = mode;
				if (!im.ec.empty()) {
					if (mode == 2 || mode == 3) { (void) br.u32(0, 0, 1, 0, 2, 0, 3, 3); (void) br.u(1); }
					else if (mode == 4) (void) br.u(1);
				}
				if (!full_frame || mode != 0) (void) br.u(2);
			}
			if (im.have_animation) {
				duration = br.u32_64(0, 0, 1, 0, 0, 8, 0, 32);
				if (im.anim_have_timecodes) (void) br.u64bits(32);
			}
			f->is_last = br.u(1);
		} else {
			f->is_last = false;
		}
		if (f->type != 1 && !f->is_last) save_as_ref = (int32_t) br.u(2);
		if (f->type == 2 || (full_frame && (f->type == 0 || f->type == 3) && blend_mode0 == 0 && (duration == 0 || save_as_ref != 0) && !f->is_last)) (void) br.u(1);
		skip_name(br);
		{   // RestorationFilter. The reference reads the conditional fields even when all_default is
			// set (j40.h:5339-5366); a drop-in has to consume the same bits.
			FrameHeader::Restoration &rf = f->restoration;
			bool all_default = br.u(1);
			rf.gab = all_default ? true : br.u(1);
			if (rf.gab && br.u(1)) for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) rf.gab_weights[i][j] = br.f16();
			rf.epf_iters = all_default ? 2 : (int32_t) br.u(2);
			if (rf.epf_iters) {
				if (!f->is_modular && br.u(1)) for (int i = 0; i < 8; ++i) rf.sharp_lut[i] = br.f16();
				if (br.u(1)) { for (int i = 0; i < 3; ++i) rf.channel_scale[i] = br.f16(); br.skip_bits_like_reference(32); }
				if (br.u(1)) { if (!f->is_modular) rf.quant_mul = br.f16(); rf.pass0_sigma_scale = br.f16(); rf.pass2_sigma_scale = br.f16(); rf.border_sad_mul = br.f16(); }
				if (f->is_modular) rf.sigma_for_modular = br.f16();
			}
			if (!all_default) read_extensions(br);
		}
		read_extensions(br);
	}
	f->grows = ceil_div(f->height, 1 << f->group_size_shift);
	f->gcolumns = ceil_div(f->width, 1 << f->group_size_shift);
	f->num_groups = (int64_t) f->grows * f->gcolumns;
	f->ggrows = ceil_div(f->height, 8 << f->group_size_shift);
	f->ggcolumns = ceil_div(f->width, 8 << f->group_size_shift);
	f->num_lf_groups = (int64_t) f->ggrows * f->ggcolumns;
}

GroupInfo group_info(const FrameHeader &fh, int64_t gidx) {  // j40.h:7734
	GroupInfo g;
	const int32_t shift = fh.group_size_shift;
	int64_t row = gidx / fh.gcolumns, column = gidx % fh.gcolumns;
	g.ggidx = (int32_t) ((row / 8) * fh.ggcolumns + column / 8);
	g.gx_in_gg = (int32_t) (column % 8) << shift;
	g.gy_in_gg = (int32_t) (row % 8) << shift;
	g.gw = (int32_t) (std::min<int64_t>(fh.width, (column + 1) << shift) - (column << shift));
	g.gh = (int32_t) (std::min<int64_t>(fh.height, (row + 1) << shift) - (row << shift));
	return g;
}

// ------------------------------------------------------------------------------------------------
// TOC (j40.h:5479). The whole codestream is in memory, so sections are addressed directly; the
// reference's dependency reordering is a streaming concern and not needed.

static void read_toc(BitReader &br, const FrameHeader &fh, Toc *toc) {
	int64_t nsections = fh.num_passes == 1 && fh.num_groups == 1 ? 1 : 1 + fh.num_lf_groups + 1 + fh.num_passes * fh.num_groups;
	J40HIP_SHOULD(nsections <= INT32_MAX, "flen");
	std::vector<int32_t> lehmer;
	if (br.u(1)) {  // permuted
		CodeSpec spec;
		read_code_spec(br, 8, &spec);
		CodeState code(&spec);
		lehmer = read_permutation(br, code, (int32_t) nsections, 0);
		finish_code(br, code);
	}
	br.zero_pad_to_byte();
	std::vector<Section> sections((size_t) nsections);
	for (Section &s : sections) s.size = (size_t) br.u32(0, 10, 1024, 14, 17408, 22, 4211712, 30);
	br.zero_pad_to_byte();
	size_t off = br.byte_position();
	for (Section &s : sections) { s.offset = off; off += s.size; }
	toc->end_offset = off;
	if (nsections == 1) { toc->single = true; toc->single_section = sections[0]; return; }
	if (!lehmer.empty()) apply_permutation(sections.data(), lehmer);
	toc->lf_global = sections[0];
	toc->lf_groups.assign(sections.begin() + 1, sections.begin() + 1 + fh.num_lf_groups);
	toc->hf_global = sections[(size_t) (1 + fh.num_lf_groups)];
	toc->pass_groups.assign(sections.begin() + 2 + fh.num_lf_groups, sections.end());
}

// ------------------------------------------------------------------------------------------------
// LfGlobal (j40.h:6257)

static void init_global_modular(Frame *f) {  // j40.h:3619
	const ImageMeta &im = f->im; const FrameHeader &fh = f->fh;
	int32_t n = (int32_t) im.ec.size();
	if (fh.is_modular) n += (!fh.do_ycbcr && !im.xyb_encoded && im.grey) ? 1 : 3;
	f->gmodular.channel.assign((size_t) n, Plane());
	for (size_t i = 0; i < im.ec.size(); ++i) J40HIP_SHOULD(im.ec[i].dim_shift == 0, "TODO");
	for (Plane &p : f->gmodular.channel) { p.width = fh.width; p.height = fh.height; }
	f->gmodular.bpp = im.bpp;
}

static void read_lf_global(BitReader &br, Frame *f) {
	const FrameHeader &fh = f->fh;
	J40HIP_SHOULD(!fh.has_patches && !fh.has_splines && !fh.has_noise, "TODO");
	if (!br.u(1)) for (int i = 0; i < 3; ++i) f->m_lf_scaled[i] = br.f16() / 128.0f;
	if (!fh.is_modular) {
		f->global_scale = br.u32(1, 11, 2049, 11, 4097, 12, 8193, 16);
		f->quant_lf = br.u32(16, 0, 1, 5, 1, 8, 1, 16);
		if (br.u(1)) {
			static const uint8_t DEFAULT_MAP[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
			f->block_ctx_map.assign(DEFAULT_MAP, DEFAULT_MAP + 39);
			f->nb_block_ctx = 15;
		} else {
			int32_t size = 39;
			for (int i = 0; i < 3; ++i) {
				f->nb_lf_thr[i] = (int32_t) br.u(4);
				for (int32_t j = 0; j < f->nb_lf_thr[i]; ++j) f->lf_thr[i][j] = (int32_t) unpack_signed64(br.u32_64(0, 4, 16, 8, 272, 16, 65808, 32));
				size *= f->nb_lf_thr[i] + 1;
			}
			f->nb_qf_thr = (int32_t) br.u(4);
			for (int32_t i = 0; i < f->nb_qf_thr; ++i) f->qf_thr[i] = br.u32(0, 2, 4, 3, 12, 5, 44, 8) + 1;
			size *= f->nb_qf_thr + 1;
			J40HIP_SHOULD(size <= 39 * 64, "hfbc");
			read_cluster_map(br, size, 16, &f->nb_block_ctx, &f->block_ctx_map);
		}
		if (!br.u(1)) {
			f->inv_colour_factor = 1.0f / (float) br.u32(84, 0, 256, 0, 2, 8, 258, 16);
			f->base_corr_x = br.f16();
			f->base_corr_b = br.f16();
			f->x_factor_lf = (int32_t) br.u(8) - 127;
			f->b_factor_lf = (int32_t) br.u(8) - 127;
		}
	}
	init_global_modular(f);
	if (br.u(1)) {  // global tree
		// tree size limit with the reference's saturating 32-bit products (j40.h:6321-6322)
		int64_t area = std::min<int64_t>((int64_t) fh.width * fh.height, INT32_MAX);
		int64_t prod = std::min<int64_t>(area * (int64_t) f->gmodular.channel.size(), INT32_MAX);
		int64_t limit = std::min<int64_t>(1 << 22, 1024 + prod / 16);
		read_tree(br, (int32_t) limit, 64, &f->global_tree, &f->global_codespec);
	}
	if (!f->gmodular.channel.empty()) {
		read_modular_header(br, &f->global_tree, &f->global_codespec, &f->gmodular);
		// LfGlobal codes the meta channels and the channels behind them that fit into one group; the rest is left to the
		// LfGroup / pass-group sections. Without Squeeze that is the reference's rule (every channel of a frame no larger
		// than a group, else the meta channels only, j40.h:6329-6333); after a Squeeze the small channels qualify too
		{
			const int32_t gdim = 1 << fh.group_size_shift, nch = (int32_t) f->gmodular.channel.size();
			int32_t n = f->gmodular.nb_meta_channels;
			while (n < nch && f->gmodular.channel[(size_t) n].width <= gdim && f->gmodular.channel[(size_t) n].height <= gdim) ++n;
			f->num_gm_channels = n;
		}
		if (!fh.is_modular) {
			// A squeezed extra-channel image of a VarDCT frame that is not finished inside LfGlobal would need the ModularLfGroup data
			// between the LF coefficients and the HF metadata of every LfGroup section and shift-aware sub-images behind the
			// coefficients of every pass group; neither exists here, and the reference stops at the Squeeze parameters (j40.h:3812)
			bool squeezed = false;
			for (const Transform &t : f->gmodular.transforms) squeezed = squeezed || t.kind == Transform::SQUEEZE;
			J40HIP_SHOULD(!squeezed || f->num_gm_channels == (int32_t) f->gmodular.channel.size(), "TODO");
			allocate_modular(&f->gmodular);   // (Modular frames are decoded on the device: no host planes)
		}
		if (fh.is_modular) {
			// the channel data that follows (j40.h:6334-6337) is decoded by the HIP Modular kernel, which
			// continues from this bit position inside the section
			f->gm_data_bitpos = br.bit_position();
			f->gm_data_pending = true;
			return;
		}
		CodeState code(f->gmodular.codespec);
		for (int32_t i = 0; i < f->num_gm_channels; ++i) decode_modular_channel(br, f->gmodular, code, i, 0);
		finish_code(br, code);
	}
}

// ------------------------------------------------------------------------------------------------
// HfGlobal + HfPass (j40.h:6819)

static void read_dq_matrix(BitReader &br, int32_t idx, int64_t raw_sidx, Frame *f) {  // j40.h:4696
	DqMatrix &dq = f->dq_matrix[idx];
	static const int8_t LOGDIM[17][2] = {{3, 3}, {3, 3}, {3, 3}, {3, 3}, {4, 4}, {5, 5}, {3, 4}, {3, 5}, {4, 5}, {3, 3}, {3, 3}, {6, 6}, {5, 6}, {7, 7}, {6, 7}, {8, 8}, {7, 8}};
	const int32_t nrows = 1 << LOGDIM[idx][0], ncols = 1 << LOGDIM[idx][1];
	dq.mode = (int32_t) br.u(3);
	dq.params.clear();
	if (dq.mode == 7) {  // raw: coded as a 3-channel Modular image
		float denom = br.f16();
		J40HIP_SHOULD(std::isfinite(denom) && std::fabs(denom) >= 1e-8f, "dqm0");
		float inv_denom = 1.0f / denom;
		Modular m;
		m.bpp = f->im.bpp;
		m.channel.assign(3, Plane());
		for (Plane &p : m.channel) { p.width = ncols; p.height = nrows; }
		decode_modular_image(br, &f->global_tree, &f->global_codespec, raw_sidx, &m);
		dq.params.assign((size_t) (nrows * ncols), std::array<float, 3>{0, 0, 0});
		for (int c = 0; c < 3; ++c) for (int32_t i = 0; i < nrows * ncols; ++i) dq.params[(size_t) i][(size_t) c] = (float) m.channel[(size_t) c].px[(size_t) i] * inv_denom;
		dq.n = nrows; dq.m = ncols;
	} else {
		static const struct { int8_t needs8x8, nparams, nscaled, ndct; } HOW[7] = {{0, 0, 0, 0}, {1, 3, 3, 0}, {1, 6, 6, 0}, {1, 2, 2, 1}, {1, 1, 0, 1}, {1, 9, 6, 2}, {1, 0, 0, 1}};
		const auto how = HOW[dq.mode];
		if (how.needs8x8) J40HIP_SHOULD(nrows == 8 && ncols == 8, "dqm?");
		int32_t paramsize = how.nparams + how.ndct * 16, at = how.nparams;
		if (paramsize) {
			dq.params.assign((size_t) paramsize, std::array<float, 3>{0, 0, 0});
			for (int c = 0; c < 3; ++c) for (int32_t j = 0; j < how.nparams; ++j) dq.params[(size_t) j][(size_t) c] = br.f16() * (j < how.nscaled ? 64.0f : 1.0f);
			for (int32_t i = 0; i < how.ndct; ++i) {
				int32_t n = (int32_t) br.u(4) + 1;
				(i == 0 ? dq.n : dq.m) = n;
				for (int c = 0; c < 3; ++c) for (int32_t j = 0; j < n; ++j) dq.params[(size_t) (at + j)][(size_t) c] = br.f16() * (j == 0 ? 64.0f : 1.0f);
				at += n;
			}
		}
	}
}

static void read_hf_global(BitReader &br, Frame *f) {
	const FrameHeader &fh = f->fh;
	const int64_t sidx_base = 1 + 3 * fh.num_lf_groups;
	if (!br.u(1)) for (int32_t i = 0; i < 17; ++i) read_dq_matrix(br, i, sidx_base + i, f);
	f->num_hf_presets = (int32_t) br.u(ceil_lg32((uint32_t) fh.num_groups)) + 1;
	for (int32_t pass = 0; pass < fh.num_passes; ++pass) {
		int32_t used_orders = br.u32(0x5f, 0, 0x13, 0, 0, 0, 0, 13);
		if (used_orders > 0) {
			CodeSpec spec;
			read_code_spec(br, 8, &spec);
			CodeState code(&spec);
			for (int32_t j = 0; j < 13; ++j) if (used_orders >> j & 1) {
				int32_t size = 1 << (LOG_ORDER_SIZE[j][0] + LOG_ORDER_SIZE[j][1]);
				for (int c = 0; c < 3; ++c) { f->order_lehmer[pass][j][c] = read_permutation(br, code, size, size / 64); f->order_has_lehmer[pass][j][c] = true; }
			}
			finish_code(br, code);
		}
		read_code_spec(br, 495 * f->nb_block_ctx * f->num_hf_presets, &f->coeff_codespec[pass]);
	}
}

// ------------------------------------------------------------------------------------------------
// LfGroup (j40.h:6492-6790)

static void smooth_lf(const Frame &f, int32_t w8, int32_t h8, std::vector<float> lfq[3]) {  // j40.h:6492
	static const float W0 = 0.05226273532324128f, W1 = 0.20345139757231578f, W2 = 0.0334829185968739f;
	float inv_m_lf[3];
	for (int c = 0; c < 3; ++c) inv_m_lf[c] = (float) (f.global_scale * f.quant_lf) / f.m_lf_scaled[c] / 65536.0f;
	if (h8 < 3 || w8 < 3) return;
	std::vector<float> prev[3], cur[3];
	for (int c = 0; c < 3; ++c) { cur[c].assign(lfq[c].begin(), lfq[c].begin() + w8); prev[c].resize((size_t) w8); }
	for (int32_t y = 1; y < h8 - 1; ++y) {
		float *out[3]; const float *south[3];
		for (int c = 0; c < 3; ++c) {
			prev[c].swap(cur[c]);
			out[c] = lfq[c].data() + (size_t) y * (size_t) w8;
			south[c] = out[c] + w8;
			cur[c].assign(out[c], out[c] + w8);  // unsmoothed copy of this row
		}
		for (int32_t x = 1; x < w8 - 1; ++x) {
			float wa[3], gap = 0.5f;
			for (int c = 0; c < 3; ++c) {
				const float *n = prev[c].data(), *l = cur[c].data(), *s = south[c];
				wa[c] = (n[x - 1] * W2 + n[x] * W1 + n[x + 1] * W2) + (l[x - 1] * W1 + l[x] * W0 + l[x + 1] * W1) + (s[x - 1] * W2 + s[x] * W1 + s[x + 1] * W2);
				float diff = fabsf(wa[c] - l[x]) * inv_m_lf[c];
				if (gap < diff) gap = diff;
			}
			gap = 3.0f - 4.0f * gap;
			gap = 0.0f > gap ? 0.0f : gap;
			for (int c = 0; c < 3; ++c) out[c][x] = (wa[c] - cur[c][(size_t) x]) * gap + cur[c][(size_t) x];
		}
	}
}

// the tail of an LfGroup: dequantisation (j40.h:6562), adaptive smoothing (j40.h:6492), LLF coefficients of every varblock
// (j40.h:6668-6683, 5944) from the decoded integers and the varblock layout. The device does the same at upload
// (device/lf_tail_kernels: k_lf_dequant_smooth, k_llf) when Frame::defer_lf_tail is set.
static void lf_tail_on_host(const Frame &f, LfGroup *gg) {
	if (!gg->tail_pending) return;
	const int32_t w8 = gg->width8, h8 = gg->height8;
	std::vector<float> lfq[3];
	for (int c = 0; c < 3; ++c) {
		lfq[c].resize((size_t) w8 * (size_t) h8);
		for (size_t i = 0; i < lfq[c].size(); ++i) lfq[c][i] = (float) gg->lfraw[c][i] * gg->mult_lf[c];
	}
	if (!f.fh.skip_adapt_lf_smooth) smooth_lf(f, w8, h8, lfq);
	for (int c = 0; c < 3; ++c) gg->llfcoeffs[c].assign((size_t) w8 * (size_t) h8, 0.0f);
	std::vector<float> scratch(1024);
	for (const VarblockInfo &vb : gg->varblocks) {
		const DctSelect &dct = DCT_SELECT[vb.dctsel];
		const int32_t vw8 = 1 << (dct.log_columns - 3), vh8 = 1 << (dct.log_rows - 3), coeffoff = vb.coeffoff_qfidx & ~15;
		for (int c = 0; c < 3; ++c) {
			float *llf = gg->llfcoeffs[c].data() + (coeffoff >> 6);
			for (int32_t i = 0; i < vh8; ++i) for (int32_t j = 0; j < vw8; ++j) llf[i * vw8 + j] = lfq[c][(size_t) (vb.y8 + i) * (size_t) w8 + (size_t) (vb.x8 + j)];
			if (vw8 > 1 || vh8 > 1) forward_dct2d_scaled_for_llf(llf, scratch.data(), dct.log_rows - 3, dct.log_columns - 3);
		}
	}
	gg->tail_pending = false;
}
void finish_lf_tail(Frame *f) { for (LfGroup &gg : f->lf_groups) if (gg.loaded) lf_tail_on_host(*f, &gg); }

// what follows the two Modular sub-images of an LfGroup section (j40.h:6562-6570, 6634-6688): dequantisation factors, LF index,
// varblock placement. lf: the LF integers in streamed order Y, X, B (moved into gg->lfraw when `take` is given, else copied)
static void lf_group_finish(Frame *f, LfGroup *gg, int32_t extra_prec, const int16_t *const lf[3], std::vector<int16_t> *take[3],
		const int16_t *xfromy, const int16_t *bfromy, const int16_t *info0, const int16_t *info1, int32_t nb_varblocks) {
	const FrameHeader &fh = f->fh;
	const int32_t w8 = gg->width8, h8 = gg->height8, w64 = gg->width64, h64 = gg->height64;
	const size_t cells = (size_t) w8 * (size_t) h8;
	static const int XYB_FROM_STREAM[3] = {1, 0, 2};
	const int16_t *ch[3];
	for (int c = 0; c < 3; ++c) {
		gg->mult_lf[c] = f->m_lf_scaled[c] / (float) (f->global_scale * f->quant_lf) * (float) (65536 >> extra_prec);  // j40.h:6562
		ch[c] = lf[XYB_FROM_STREAM[c]];
	}
	// LF index: thresholds counted on the raw integers; note each factor is a channel's own threshold count (j40.h:6566-6570)
	gg->lfindices.assign(cells, 0);
	auto add = [&](const int16_t *p, const int32_t *thr, int32_t n) { for (int32_t t = 0; t < n; ++t) for (size_t i = 0; i < cells; ++i) gg->lfindices[i] = (uint8_t) (gg->lfindices[i] + (p[i] > thr[t])); };
	auto mul = [&](int32_t k) { if (k != 1) for (uint8_t &v : gg->lfindices) v = (uint8_t) (v * k); };
	add(ch[0], f->lf_thr[0], f->nb_lf_thr[0]); mul(f->nb_lf_thr[0] + 1);
	add(ch[2], f->lf_thr[2], f->nb_lf_thr[2]); mul(f->nb_lf_thr[2] + 1);
	add(ch[1], f->lf_thr[1], f->nb_lf_thr[1]);
	for (int c = 0; c < 3; ++c) {
		if (take) gg->lfraw[c].swap(*take[XYB_FROM_STREAM[c]]);
		else gg->lfraw[c].assign(ch[c], ch[c] + cells);
	}
	gg->tail_pending = true;
	gg->xfromy.assign(xfromy, xfromy + (size_t) w64 * (size_t) h64); gg->bfromy.assign(bfromy, bfromy + (size_t) w64 * (size_t) h64);

	// place varblocks in raster order at the first free cell (j40.h:6634-6688)
	const int32_t log_gsize8 = fh.group_size_shift - 3;
	gg->blocks.assign(cells, 0);
	gg->varblocks.clear(); gg->varblocks.reserve((size_t) nb_varblocks);   // (filled in placement order below: no need to zero 20 bytes per block first)
	int32_t voff = 0, coeffoff = 0;
	uint32_t dct_used = 0, order_used = 0;
	for (int32_t y0 = 0; y0 < h8; ++y0) for (int32_t x0 = 0; x0 < w8; ++x0) {
		if (gg->blocks[(size_t) y0 * (size_t) w8 + (size_t) x0]) continue;
		J40HIP_SHOULD(voff < nb_varblocks, "vblk");
		const int32_t dctsel = info0[voff];
		J40HIP_SHOULD(0 <= dctsel && dctsel < 27, "dct?");
		const DctSelect &dct = DCT_SELECT[dctsel];
		dct_used |= 1u << dctsel; order_used |= 1u << dct.order_idx;
		const int32_t vw8 = 1 << (dct.log_columns - 3), vh8 = 1 << (dct.log_rows - 3);
		const int32_t x1 = x0 + vw8 - 1, y1 = y0 + vh8 - 1;
		J40HIP_SHOULD(x1 < w8 && (x0 >> log_gsize8) == (x1 >> log_gsize8), "vblk");
		J40HIP_SHOULD(y1 < h8 && (y0 >> log_gsize8) == (y1 >> log_gsize8), "vblk");
		// (the reference does not check this, see its note at j40.h:6691: blocks overlapping earlier ones can add up to more cells
		// than the LfGroup has, and it then writes past its coefficient arrays; same check as device/plan_dev.h)
		J40HIP_SHOULD((size_t) coeffoff + ((size_t) 1 << (dct.log_columns + dct.log_rows)) <= cells * 64, "vblk");
		for (int32_t i = 0; i < vh8; ++i) for (int32_t j = 0; j < vw8; ++j) gg->blocks[(size_t) (y0 + i) * (size_t) w8 + (size_t) (x0 + j)] = 1 << 20 | voff;
		gg->blocks[(size_t) y0 * (size_t) w8 + (size_t) x0] = (dctsel + 2) << 20 | voff;
		gg->varblocks.push_back(VarblockInfo());
		VarblockInfo &vb = gg->varblocks.back();
		vb.coeffoff_qfidx = coeffoff; vb.x8 = x0; vb.y8 = y0; vb.dctsel = dctsel;
		const int32_t hfmul_m1 = info1[voff];
		for (int32_t j = 0; j < f->nb_qf_thr; ++j) vb.coeffoff_qfidx += hfmul_m1 >= f->qf_thr[j];
		vb.hfmul_inv = 1.0f / ((float) hfmul_m1 + 1.0f);
		coeffoff += 1 << (dct.log_columns + dct.log_rows);
		++voff;
	}
	J40HIP_SHOULD(voff == nb_varblocks, "vblk");
	gg->loaded = true;
	if (!f->defer_lf_tail) lf_tail_on_host(*f, gg);
	static std::mutex mu;
	std::lock_guard<std::mutex> lock(mu);
	f->dct_select_used |= dct_used; f->order_used |= order_used;
}

// the streams of an LfGroup section (j40.h:6722-6790) as decoded, nothing derived yet
void read_lf_group_raw(BitReader &br, const Frame &f, const LfGroup &gg, LfRaw *out) {
	const FrameHeader &fh = f.fh;
	const int64_t sidx0 = 1 + gg.idx, sidx2 = 1 + 2 * fh.num_lf_groups + gg.idx;
	const int32_t w8 = gg.width8, h8 = gg.height8, w64 = gg.width64, h64 = gg.height64;
	J40HIP_SHOULD(!fh.use_lf_frame, "TODO");
	J40HIP_SHOULD(fh.jpeg_upsampling == 0, "TODO");

	// LF image: three channels in Y, X, B order
	out->extra_prec = (int32_t) br.u(2);
	Modular lfm; lfm.bpp = f.im.bpp;
	lfm.channel.assign(3, Plane());
	for (Plane &p : lfm.channel) { p.width = w8; p.height = h8; }
	decode_modular_image(br, &f.global_tree, &f.global_codespec, sidx0, &lfm);
	for (int c = 0; c < 3; ++c) J40HIP_SHOULD(lfm.channel[(size_t) c].width == w8 && lfm.channel[(size_t) c].height == h8, "TODO");

	// HF metadata
	const int32_t nb_varblocks = (int32_t) br.u(ceil_lg32((uint32_t) (w8 * h8))) + 1;
	Modular m; m.bpp = f.im.bpp;
	m.channel.assign(4, Plane());
	m.channel[0].width = m.channel[1].width = w64; m.channel[0].height = m.channel[1].height = h64;
	m.channel[2].width = nb_varblocks; m.channel[2].height = 2;
	m.channel[3].width = w8; m.channel[3].height = h8;
	decode_modular_image(br, &f.global_tree, &f.global_codespec, sidx2, &m);
	J40HIP_SHOULD(m.channel.size() == 4 && m.channel[2].width == nb_varblocks && m.channel[2].height == 2, "TODO");
	J40HIP_SHOULD((int32_t) m.channel[0].px.size() == w64 * h64 && (int32_t) m.channel[1].px.size() == w64 * h64, "TODO");
	out->nb_varblocks = nb_varblocks;
	for (int c = 0; c < 3; ++c) out->lf[c].swap(lfm.channel[(size_t) c].px);
	out->xfromy.swap(m.channel[0].px); out->bfromy.swap(m.channel[1].px); out->info.swap(m.channel[2].px); out->sharp.swap(m.channel[3].px);
}

static void read_lf_group(BitReader &br, Frame *f, LfGroup *gg) {  // j40.h:6722
	if (f->fh.is_modular) { gg->loaded = true; return; }  // nothing is read: no channel has shift >= 3 (j40.h:6731)
	LfRaw raw;
	read_lf_group_raw(br, *f, *gg, &raw);
	const int16_t *lf[3] = {raw.lf[0].data(), raw.lf[1].data(), raw.lf[2].data()};
	std::vector<int16_t> *take[3] = {&raw.lf[0], &raw.lf[1], &raw.lf[2]};
	lf_group_finish(f, gg, raw.extra_prec, lf, take, raw.xfromy.data(), raw.bfromy.data(), raw.info.data(), raw.info.data() + raw.nb_varblocks, raw.nb_varblocks);
	gg->sharpness.swap(raw.sharp);
}

static void allocate_lf_groups(Frame *f) {  // j40.h:7659
	const FrameHeader &fh = f->fh;
	const int32_t ggsize = 8 << fh.group_size_shift;
	f->lf_groups.assign((size_t) fh.num_lf_groups, LfGroup());
	int32_t idx = 0;
	for (int32_t ggy = 0; ggy < fh.height; ggy += ggsize) for (int32_t ggx = 0; ggx < fh.width; ggx += ggsize, ++idx) {
		LfGroup &gg = f->lf_groups[(size_t) idx];
		gg.idx = idx; gg.left = ggx; gg.top = ggy;
		gg.width = std::min(ggsize, fh.width - ggx); gg.height = std::min(ggsize, fh.height - ggy);
		gg.width8 = ceil_div(gg.width, 8); gg.height8 = ceil_div(gg.height, 8);
		gg.width64 = ceil_div(gg.width, 64); gg.height64 = ceil_div(gg.height, 64);
	}
}

static void prepare_tables(Frame *f) {  // j40.h:7694-7732
	for (int32_t i = 0; i < 27; ++i) if (f->dct_select_used >> i & 1) load_dq_matrix(DCT_SELECT[i].param_idx, &f->dq_matrix[DCT_SELECT[i].param_idx]);
	for (int32_t i = 0; i < 13; ++i) if (f->order_used >> i & 1) {
		const int32_t skip = 1 << (LOG_ORDER_SIZE[i][0] + LOG_ORDER_SIZE[i][1] - 6);
		for (int32_t pass = 0; pass < f->fh.num_passes; ++pass) for (int c = 0; c < 3; ++c) {
			std::vector<int32_t> &order = f->orders[pass][i][c];
			if (!order.empty()) continue;
			natural_order(LOG_ORDER_SIZE[i][0], LOG_ORDER_SIZE[i][1], &order);
			if (f->order_has_lehmer[pass][i][c]) apply_permutation(order.data() + skip, f->order_lehmer[pass][i][c]);
		}
	}
}

// ------------------------------------------------------------------------------------------------

// ICC stream (j40.h:3351-3393): like the reference, decoded and discarded -- the renderer always produces sRGB
static void skip_icc(BitReader &br) {
	const uint64_t enc_size = br.u64();
	CodeSpec spec;
	read_code_spec(br, 41, &spec);
	CodeState code(&spec);
	uint64_t index = 0, output_size = 0;
	{   // output size: a varint of bytes coded with context 0
		int32_t shift = 0;
		for (;;) {
			J40HIP_SHOULD(index++ < enc_size, "icc?");
			const int32_t b = decode_symbol(br, code, 0, 0);
			output_size |= (uint64_t) (b & 0x7f) << shift;
			if (b < 128) break;
			shift += 7;
			J40HIP_SHOULD(shift < 63, "vint");
		}
	}
	J40HIP_SHOULD(output_size <= ((uint64_t) 1 << 22), "plim");   // main profile, level 5 (j40.h:1172)
	J40HIP_SHOULD(output_size >= enc_size / 21, "icc?");
	int32_t byte = 0, prev = 0, pprev = 0;
	auto kind = [](int32_t v) { return (97 <= (v | 32) && (v | 32) <= 122) ? 0 : (v == 44 || v == 46 || (48 <= v && v < 58)) ? 1 : 4; };
	for (; index < enc_size; ++index) {
		pprev = prev; prev = byte;
		int32_t ctx = 0;
		if (index > 128) {
			if (prev < 16) ctx = prev < 2 ? prev + 3 : 5;
			else if (prev > 240) ctx = 6 + (prev == 255);
			else ctx = kind(prev) == 0 ? 1 : kind(prev) == 1 ? 2 : 8;
			ctx += 8 * (pprev < 16 ? 2 : pprev > 240 ? 3 : kind(pprev));
		}
		byte = decode_symbol(br, code, ctx, 0);
	}
	finish_code(br, code);
}

// signature, image and frame headers, TOC (sections clipped to the bytes that exist), the LfGroups' geometry
// `limit`: how much of the codestream's cs_size bytes may be read (streaming input: what has arrived; else cs_size)
static void parse_headers_within(const uint8_t *cs, size_t limit, size_t cs_size, Frame *f) {
	memset(f->order_has_lehmer, 0, sizeof f->order_has_lehmer);
	BitReader br(cs, limit);
	J40HIP_SHOULD(br.u(16) == 0x0aff, "!jxl");
	read_image_metadata(br, &f->im);
	if (f->im.want_icc) skip_icc(br);
	read_frame_header(br, f->im, &f->fh);
	J40HIP_SHOULD(f->fh.is_last, "TODO");
	J40HIP_SHOULD(f->fh.type == 0, "TODO");
	read_toc(br, f->fh, &f->toc);
	{   // a truncated codestream fails where the reference fails: in the first section (in reading order) that
		// needs the missing bytes, not up front. Sections are clipped to the bytes that exist; readers raise "shrt".
		auto clip = [cs_size](Section &s) { if (s.offset >= cs_size) { s.offset = cs_size; s.size = 0; } else s.size = std::min(s.size, cs_size - s.offset); };
		if (f->toc.single) {
			f->toc.single_declared_end = f->toc.single_section.offset + f->toc.single_section.size;
			f->toc.single_section.size = cs_size > f->toc.single_section.offset ? cs_size - f->toc.single_section.offset : 0;
		}
		clip(f->toc.single_section); clip(f->toc.lf_global); clip(f->toc.hf_global);
		for (Section &s : f->toc.lf_groups) clip(s);
		for (Section &s : f->toc.pass_groups) clip(s);
	}
	allocate_lf_groups(f);
}
// Headers and TOC. With a streaming source (Frame::need_bytes) their extent is not known ahead: they are parsed on the prefix that
// has arrived and, when that runs out ("shrt": every bit read until then was real, so any other error is the stream's own), again
// on a longer one -- a few hundred bytes for most streams, an embedded ICC profile's worth for some.
static void parse_headers(const uint8_t *cs, size_t cs_size, Frame *f) {
	if (!f->need_bytes || !f->have_bytes) { parse_headers_within(cs, cs_size, cs_size, f); return; }
	size_t want = std::min<size_t>(cs_size, 4096);
	for (;;) {
		f->need(want);
		const size_t have = std::min(cs_size, std::max(want, f->have_bytes(f->need_ctx)));
		try { parse_headers_within(cs, have, cs_size, f); return; }
		catch (const DecodeError &e) {
			if (e.code != (uint32_t) E4("shrt") || have >= cs_size) throw;
			*f = f->with_same_inputs();
			want = std::min(cs_size, std::max(have * 2, have + 4096));
		}
	}
}

// LfGlobal and HfGlobal of a frame with several sections
static void parse_globals(const uint8_t *cs, Frame *f) {
	f->need(std::max(f->toc.lf_global.offset + f->toc.lf_global.size, f->toc.hf_global.offset + f->toc.hf_global.size));
	{
		BitReader sr(cs + f->toc.lf_global.offset, f->toc.lf_global.size);
		read_lf_global(sr, f);
		// (no check that the section ends here, in none of the sections of a frame that has several: the reference's
		// j40__finish_section_state runs j40__no_more_bytes on the section's own state and returns the parent's, j40.h:7778-7795)
	}
	if (f->fh.is_modular) {
		J40HIP_SHOULD(f->toc.hf_global.size == 0, "excs");
	} else {
		BitReader sr(cs + f->toc.hf_global.offset, f->toc.hf_global.size);
		read_hf_global(sr, f);
	}
}

// What the pipeline's host stage reads of a VarDCT frame with several sections (device/async.hip): everything in front of the
// LfGroup sections, and of each of those what precedes its first stream. Returns false -- nothing is lost, parse_frame does it
// all again -- for frames that are not of that kind. `tasks` (one per LfGroup section): where the streams start; `plain`: whether
// every section's first Modular header is the plain one k_lf_groups handles (global tree, no transforms).
bool parse_frame_front(const uint8_t *cs, size_t cs_size, Frame *f, std::vector<LfDeviceTask> *tasks, std::vector<int32_t> *extra_prec, bool *plain) {
	parse_headers(cs, cs_size, f);
	if (f->toc.single || f->fh.is_modular) return false;
	parse_globals(cs, f);
	if (f->fh.use_lf_frame || f->fh.jpeg_upsampling != 0) return false;
	const int64_t n = f->fh.num_lf_groups;
	tasks->assign((size_t) n, LfDeviceTask()); extra_prec->assign((size_t) n, 0);
	*plain = true;
	for (int64_t i = 0; i < n; ++i) {
		const LfGroup &gg = f->lf_groups[(size_t) i];
		LfDeviceTask &t = (*tasks)[(size_t) i];
		t.byte_off = f->toc.lf_groups[(size_t) i].offset; t.size = f->toc.lf_groups[(size_t) i].size; t.bit_off = 0;
		t.w8 = gg.width8; t.h8 = gg.height8; t.w64 = gg.width64; t.h64 = gg.height64;
		t.sidx0 = (int32_t) (1 + gg.idx); t.sidx2 = (int32_t) (1 + 2 * f->fh.num_lf_groups + gg.idx);
		t.nbvb_bits = ceil_lg32((uint32_t) (gg.width8 * gg.height8));
		try {
			BitReader sr(cs + t.byte_off, t.size);
			(*extra_prec)[(size_t) i] = (int32_t) sr.u(2);
			Modular m; m.bpp = f->im.bpp;
			m.channel.assign(3, Plane());
			for (Plane &p : m.channel) { p.width = gg.width8; p.height = gg.height8; }
			read_modular_header(sr, &f->global_tree, &f->global_codespec, &m);
			if (!(m.use_global_tree && m.transforms.empty() && m.channel.size() == 3)) *plain = false;
			t.bit_off = (uint32_t) sr.bit_position();
		} catch (const DecodeError &) { *plain = false; }   // (the section's own decode reports it in its place)
	}
	return true;
}

void parse_frame(const uint8_t *cs, size_t cs_size, Frame *f, int threads) {
	static const bool timing = getenv("J40HIP_API_TIMING") != nullptr;   // (where a parse's time goes)
	auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tp0 = timing ? now() : 0;
	parse_headers(cs, cs_size, f);
	const double tp1 = timing ? now() : 0;

	if (f->toc.single) {
		// one section holds LfGlobal, HfGlobal, LfGroup and PassGroup back to back, read in the order
		// the reference reads them (j40.h:8189-8199)
		f->need(cs_size);
		BitReader sr(cs + f->toc.single_section.offset, f->toc.single_section.size);
		read_lf_global(sr, f);
		if (f->fh.is_modular) return;   // LfGroup / PassGroup read nothing for single-group Modular frames (j40.h:6731, 7024)
		read_hf_global(sr, f);
		read_lf_group(sr, f, &f->lf_groups[0]);
		if (!f->fh.is_modular) {
			// the pass group continues in the middle of this section: the device reader starts at a bit offset
			prepare_tables(f);
			f->single_pass_group_bitpos = sr.bit_position();
			return;
		}
		// Modular: every channel was already decoded in LfGlobal (num_gm_channels = all), so the pass
		// group reads nothing (j40.h:7024-7025) and the section has to end here
		J40HIP_SHOULD(f->num_gm_channels == (int32_t) f->gmodular.channel.size(), "TODO");
		sr.zero_pad_to_byte();
		{   // j40__end_of_frame, j40.h:7796-7803: short of the TOC entry's end is `shrt`, beyond it `excs`
			const size_t at = f->toc.single_section.offset + sr.byte_position();
			J40HIP_SHOULD(at >= f->toc.single_declared_end, "shrt");
			J40HIP_SHOULD(at == f->toc.single_declared_end, "excs");
		}
		return;
	}

	parse_globals(cs, f);
	const double tp2 = timing ? now() : 0;
	// LfGroup sections are independent of each other
	const int64_t n = f->fh.num_lf_groups;
	std::atomic<int64_t> next(0);
	std::atomic<uint32_t> first_err(0);
	std::vector<uint32_t> errs((size_t) n, 0);
	std::vector<char> done((size_t) n, 0);
	if (f->lf_decoder) f->need(cs_size);   // (the device reads the sections out of the whole codestream)
	if (f->lf_decoder && !f->fh.is_modular && !f->fh.use_lf_frame && f->fh.jpeg_upsampling == 0 && f->defer_lf_tail) {
		// the streams of every LfGroup section on the device (device/lf_decode.hip): the host reads what precedes the LF coefficient
		// stream -- extra precision and the first Modular header, which has to be the plain one --, the device decodes both
		// sub-images, the host finishes (LF index, varblock placement). Anything irregular: the host path below, unchanged.
		std::vector<LfDeviceTask> tasks((size_t) n);
		std::vector<int32_t> extra_prec((size_t) n, 0);
		bool plain = true;
		try {
			for (int64_t i = 0; i < n && plain; ++i) {
				const LfGroup &gg = f->lf_groups[(size_t) i];
				BitReader sr(cs + f->toc.lf_groups[(size_t) i].offset, f->toc.lf_groups[(size_t) i].size);
				extra_prec[(size_t) i] = (int32_t) sr.u(2);
				Modular m; m.bpp = f->im.bpp;
				m.channel.assign(3, Plane());
				for (Plane &p : m.channel) { p.width = gg.width8; p.height = gg.height8; }
				read_modular_header(sr, &f->global_tree, &f->global_codespec, &m);
				plain = m.use_global_tree && m.transforms.empty() && m.channel.size() == 3;
				LfDeviceTask &t = tasks[(size_t) i];
				t.byte_off = f->toc.lf_groups[(size_t) i].offset; t.size = f->toc.lf_groups[(size_t) i].size; t.bit_off = (uint32_t) sr.bit_position();
				t.w8 = gg.width8; t.h8 = gg.height8; t.w64 = gg.width64; t.h64 = gg.height64;
				t.sidx0 = (int32_t) (1 + gg.idx); t.sidx2 = (int32_t) (1 + 2 * f->fh.num_lf_groups + gg.idx);
				t.nbvb_bits = ceil_lg32((uint32_t) (gg.width8 * gg.height8));
			}
		} catch (const DecodeError &) { plain = false; }
		if (plain && f->lf_decoder(f->lf_decoder_ctx, *f, cs, cs_size, tasks)) {
			f->lf_decoded_on_device = true;
			for (int64_t i = 0; i < n; ++i) {
				const LfDeviceTask &t = tasks[(size_t) i];
				if (t.status == (uint32_t) E4("lffb")) continue;   // the host decodes this one
				done[(size_t) i] = 1;
				if (t.status) { errs[(size_t) i] = t.status; continue; }
				try {
					lf_group_finish(f, &f->lf_groups[(size_t) i], extra_prec[(size_t) i], t.lf, nullptr, t.xfromy, t.bfromy, t.info0, t.info1, t.nb_varblocks);
					if (t.sharp) f->lf_groups[(size_t) i].sharpness.assign(t.sharp, t.sharp + (size_t) t.w8 * (size_t) t.h8);
				}
				catch (const DecodeError &e) { errs[(size_t) i] = e.code; }
				catch (const std::exception &) { errs[(size_t) i] = E4("!mem"); }
			}
		}
	}
	auto worker = [&]() {
		for (;;) {
			int64_t i = next.fetch_add(1);
			if (i >= n) break;
			if (done[(size_t) i]) continue;
			try {
				f->need(f->toc.lf_groups[(size_t) i].offset + f->toc.lf_groups[(size_t) i].size);   // (streaming input: this section's bytes)
				BitReader sr(cs + f->toc.lf_groups[(size_t) i].offset, f->toc.lf_groups[(size_t) i].size);
				read_lf_group(sr, f, &f->lf_groups[(size_t) i]);
			} catch (const DecodeError &e) { errs[(size_t) i] = e.code; }
			catch (const std::exception &) { errs[(size_t) i] = E4("!mem"); }   // (an exception leaving a thread would end the process)
		}
	};
	int nthreads = (int) std::min<int64_t>(threads < 1 ? 1 : threads, n);
	if (nthreads <= 1) worker();
	else {
		std::vector<std::thread> pool;
		for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);   // (the calling thread is one of the team)
		worker();
		for (auto &t : pool) t.join();
	}
	{   // the first failing section in the order the reference reads them: by offset (a permuted TOC stores them out of index order, j40.h:5608)
		size_t best = SIZE_MAX; uint32_t code = 0;
		for (size_t i = 0; i < errs.size(); ++i) if (errs[i] && f->toc.lf_groups[i].offset < best) { best = f->toc.lf_groups[i].offset; code = errs[i]; }
		if (code) raise(code);
	}
	(void) first_err;
	const double tp3 = timing ? now() : 0;
	if (!f->fh.is_modular) prepare_tables(f);
	if (timing) fprintf(stderr, "[j40hip parse] headers + TOC %.2f ms, LfGlobal + HfGlobal %.2f ms, %lld LfGroups on %d threads %.2f ms, tables %.2f ms\n", tp1 - tp0, tp2 - tp1, (long long) n, nthreads, tp3 - tp2, now() - tp3);
}

} // namespace j40hip
