// j40_amd/csrc/device/entropy_dev.h -- per-lane sequential entropy decoder for the HIP kernels:
// LSB-first bit reader, rANS (alias table), prefix codes, hybrid integers, LZ77 (one sequential
// bitstream per wavefront lane; hundreds of independent sections per launch).
//
// Reference behaviour: j40__always_refill / j40__u (j40.h:1847, 1914), j40__ans_code (2441),
// j40__prefix_code (2256), j40__hybrid_int (2313), j40__code (2804), j40__finish_and_free_code (2884).
//
// The functions are plain sequential code; when compiled without hipcc (tests/hostsim) the
// qualifiers vanish so the very same source can be single-stepped on the CPU against the oracle.
#pragma once
#include "plan.h"

#ifndef J40_DEV
#ifdef __HIPCC__
#define J40_DEV __device__ __forceinline__
#define J40_DEVM __device__ __forceinline__
#else
#define J40_DEV static inline
#define J40_DEVM inline
#endif
#endif

// Address-space qualifiers. A pointer the compiler cannot trace back to a kernel argument or a __shared__
// object is "generic" and costs flat_load / flat_store (slower, and they occupy both the LDS and the
// vector-memory counters). Pointers that are known to be global / LDS are declared as such.
#ifdef __HIPCC__
#define J40_GLOBAL __attribute__((address_space(1)))
#define J40_LDS __attribute__((address_space(3)))
#else
#define J40_GLOBAL
#define J40_LDS
#endif

namespace j40hip {

// In the latency-oriented launch every lane of a wavefront decodes the SAME section, so all decoder
// state is wave-uniform. Values that come back from memory land in vector registers; pulling them
// into scalar registers (v_readfirstlane) lets the compiler keep the whole serial decoder on the
// scalar unit: scalar branches instead of exec-mask juggling, 1-cycle SALU ops instead of dependent
// VALU ops. UNI = false is the throughput-oriented mode (one section per lane, divergent).
template <bool UNI> J40_DEV uint32_t uni(uint32_t v) {
#ifdef __HIPCC__
	if (UNI) return (uint32_t) __builtin_amdgcn_readfirstlane((int) v);
#endif
	return v;
}
template <bool UNI> J40_DEV int32_t uni(int32_t v) { return (int32_t) uni<UNI>((uint32_t) v); }
template <bool UNI> J40_DEV uint64_t uni64(uint64_t v) { return (uint64_t) uni<UNI>((uint32_t) v) | ((uint64_t) uni<UNI>((uint32_t) (v >> 32)) << 32); }

struct DevBits {
	const J40_GLOBAL uint8_t *base;   // start of the codestream buffer (HBM): 4-byte aligned, padded with >= 8 readable bytes
	uint32_t pos, end;     // next unread byte / end of the section, relative to base
	uint64_t bits;
	int32_t nbits;
	uint32_t err;          // first error (sticky)
	uint32_t ahead;        // the aligned 32-bit word at `pos`, loaded one refill ahead of its use
};

J40_DEV void bits_set_error(DevBits &b, uint32_t e) { if (!b.err) b.err = e; }

J40_DEV uint32_t bits_load32(const J40_GLOBAL uint8_t *p) { return *(const J40_GLOBAL uint32_t *) p; }

template <bool UNI> J40_DEV void bits_init(DevBits &b, const uint8_t *base, uint32_t byte_off, uint32_t size, uint32_t bit_off) {
	b.base = (const J40_GLOBAL uint8_t *) base; b.pos = byte_off + (bit_off >> 3); b.end = byte_off + size; b.bits = 0; b.nbits = 0; b.err = 0;
	const uint32_t rem = bit_off & 7;
	if (rem) {  // start in the middle of a byte (single-section frames)
		if (b.pos < b.end) { b.bits = (uint64_t) (uni<UNI>((uint32_t) b.base[b.pos++]) >> rem); b.nbits = 8 - (int32_t) rem; }
		else bits_set_error(b, ERR_SHRT);
	}
	while ((b.pos & 3) && b.pos < b.end) { b.bits |= (uint64_t) uni<UNI>((uint32_t) b.base[b.pos++]) << b.nbits; b.nbits += 8; }  // reach word alignment
	b.ahead = bits_load32(b.base + (b.pos & ~3u));  // inside the padded buffer even when pos == end; made uniform where it is consumed
}

// tops the accumulator up to >= 32 valid bits (as long as the section has bytes left); bytes past the
// section end are never consumed. The word consumed here was requested at the previous refill, so
// its memory latency overlaps with decoding instead of stalling this lane.
template <bool UNI> J40_DEV void bits_refill(DevBits &b) {
	if (b.nbits > 32) return;
	const uint32_t avail = b.end - b.pos;
	if (avail >= 4) {
		const uint32_t w = uni<UNI>(b.ahead);   // (the readfirstlane sits here, at the use: next to the load it would wait for the load)
		b.pos += 4;
		b.ahead = bits_load32(b.base + b.pos);
		b.bits |= (uint64_t) w << b.nbits;
		b.nbits += 32;
	} else if (avail) {
		b.bits |= (uint64_t) (uni<UNI>(b.ahead) & ((1u << (8 * avail)) - 1)) << b.nbits;
		b.nbits += 8 * (int32_t) avail; b.pos += avail;
	}
}

template <bool UNI> J40_DEV uint32_t bits_u(DevBits &b, int32_t n) {  // n in [0, 31]
	if (b.nbits < n) {
		bits_refill<UNI>(b);
		if (b.nbits < n) { bits_set_error(b, ERR_SHRT); b.bits = 0; b.nbits = 0; return 0; }
	}
	const uint32_t v = (uint32_t) b.bits & ((1u << n) - 1);
	b.bits >>= n; b.nbits -= n;
	return v;
}

// at least 16 bits visible if the section has them; missing bits read as zero (prefix codes at the
// very end of a section, j40.h:2258-2261)
template <bool UNI> J40_DEV uint32_t bits_peek16(DevBits &b) { if (b.nbits < 16) bits_refill<UNI>(b); return (uint32_t) b.bits & 0xffff; }
J40_DEV void bits_consume(DevBits &b, int32_t n) {
	if (n > b.nbits) { bits_set_error(b, ERR_SHRT); b.bits = 0; b.nbits = 0; return; }
	b.bits >>= n; b.nbits -= n;
}

// the section must end exactly here: zero padding up to the byte boundary, then no byte left
// (j40__no_more_bytes, j40.h:2011)
// (only called for frames that are a single section, see DevFrame::check_section_end: zero padding, j40.h:8203; bytes of the
// section left unread are `shrt`, j40__end_of_frame, j40.h:7796-7803)
J40_DEV void bits_finish_section(DevBits &b, uint32_t declared_end) {
	int32_t n = b.nbits & 7;
	if ((uint32_t) b.bits & ((1u << n) - 1)) bits_set_error(b, ERR_PAD0);
	b.bits >>= n; b.nbits -= n;
	const uint32_t at = b.pos - (uint32_t) (b.nbits >> 3);   // first byte not consumed
	if (at < declared_end) bits_set_error(b, ERR_SHRT);
	else if (at > declared_end) bits_set_error(b, ERR_EXCS);
}

// ------------------------------------------------------------------------------------------------

struct DevCode {
	// the code spec's scalars, copied once so that the per-symbol path never goes back to memory for them
	int32_t use_prefix_code, lz77_enabled, min_symbol, min_length, num_dist;
	uint32_t lz_len_cfg; int32_t lz_len_max_token;
	// tables; may point into LDS (staged by the kernel) or HBM
	const DevCluster *clusters;      // this spec's clusters
	const uint8_t *cluster_map;      // this spec's context -> cluster map
	const uint64_t *alias;           // base that DevCluster::table_off indexes (ANS)
	const int32_t *prefix;           // base that DevCluster::table_off indexes (prefix codes)
	uint32_t ans_state;
	int32_t log_bucket;
	// LZ77
	int32_t num_to_copy, copy_pos, num_decoded;
	int32_t *window;                 // nullptr: LZ77 unavailable
};

J40_DEV void code_init(DevCode &c, const DevCodeSpec &spec, const DevCluster *clusters, const uint8_t *cluster_map, const uint64_t *alias, const int32_t *prefix, int32_t *window) {
	c.use_prefix_code = spec.use_prefix_code; c.lz77_enabled = spec.lz77_enabled; c.min_symbol = spec.min_symbol; c.min_length = spec.min_length;
	c.num_dist = spec.num_dist; c.lz_len_cfg = spec.lz_len_cfg; c.lz_len_max_token = spec.lz_len_max_token;
	c.clusters = clusters; c.cluster_map = cluster_map; c.alias = alias; c.prefix = prefix;
	c.ans_state = 0; c.log_bucket = 12 - spec.log_alpha_size;
	c.num_to_copy = c.copy_pos = c.num_decoded = 0; c.window = window;
}

template <bool UNI> J40_DEV int32_t hybrid_int_dev(DevBits &b, int32_t token, uint32_t cfg, int32_t max_token) {  // j40.h:2313
	const int32_t split_exp = (int32_t) (cfg & 15), msb = (int32_t) ((cfg >> 4) & 15), lsb = (int32_t) ((cfg >> 8) & 15);
	const int32_t split = 1 << split_exp;
	if (token < split) return token;
	if (token > max_token) { token = max_token; bits_set_error(b, ERR_IOVF); }
	const int32_t in_token = msb + lsb;
	const int32_t midbits = split_exp - in_token + ((token - split) >> in_token);
	const int32_t mid = (int32_t) bits_u<UNI>(b, midbits);
	const int32_t top = 1 << msb;
	const int32_t lo = token & ((1 << lsb) - 1), hi = (token >> lsb) & (top - 1);
	return ((top | hi) << (midbits + lsb)) | ((mid << lsb) | lo);
}

template <bool UNI> J40_DEV int32_t ans_symbol(DevBits &b, DevCode &c, uint32_t table_off) {  // j40.h:2441
	if (c.ans_state == 0) { c.ans_state = bits_u<UNI>(b, 16); c.ans_state |= bits_u<UNI>(b, 16) << 16; }
	const uint32_t idx = c.ans_state & 0xfff, i = idx >> c.log_bucket, pos = idx & ((1u << c.log_bucket) - 1);
	const uint64_t e = uni64<UNI>(c.alias[table_off + i]);
	const bool aliased = pos >= (uint32_t) (e & 0xff);
	const uint32_t symbol = aliased ? (uint32_t) (e >> 20) & 0xff : i;
	const uint32_t offset = aliased ? (uint32_t) (e >> 8) & 0xfff : 0;
	const uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (uint32_t) (e >> 41) & 0x1fff;
	c.ans_state = d * (c.ans_state >> 12) + offset + pos;
	if (c.ans_state < (1u << 16)) c.ans_state = (c.ans_state << 16) | bits_u<UNI>(b, 16);
	return (int32_t) symbol;
}

template <bool UNI> J40_DEV int32_t prefix_symbol(DevBits &b, const DevCode &c, uint32_t table_off, int32_t fast_len, int32_t max_len) {  // j40.h:2256
	const int32_t *table = c.prefix + table_off;
	const uint32_t window = bits_peek16<UNI>(b);
	int32_t entry = uni<UNI>(table[window & ((1u << fast_len) - 1)]);
	int32_t used = 0;
	if (entry < 0 && fast_len < max_len) {
		const int32_t *ovf = table - entry;
		const uint32_t rest = window >> fast_len;
		int32_t code_len, guard = 0;
		do { entry = uni<UNI>(*ovf++); code_len = entry & 15; } while ((uint32_t) ((entry >> 4) & 0xfff) != (rest & ((1u << code_len) - 1)) && ++guard < 32768);
		used = fast_len;
	}
	bits_consume(b, used + (entry & 15));
	return entry >> 16;
}

// a cluster descriptor pulled into (scalar) registers
struct ClusterRegs { uint32_t cfg; int32_t max_token; uint32_t table_off; int32_t fast_len, max_len; };
template <bool UNI> J40_DEV ClusterRegs load_cluster(const DevCode &c, int32_t ctx) {
	const uint32_t cl = uni<UNI>((uint32_t) c.cluster_map[ctx]);
	const uint32_t *p = (const uint32_t *) (c.clusters + cl);   // {cfg, max_token, table_off, fast_len | max_len << 16}
	ClusterRegs r;
	r.cfg = uni<UNI>(p[0]); r.max_token = (int32_t) uni<UNI>(p[1]); r.table_off = uni<UNI>(p[2]);
	r.fast_len = r.max_len = 0;
	if (c.use_prefix_code) { const uint32_t fm = uni<UNI>(p[3]); r.fast_len = (int32_t) (int16_t) (fm & 0xffff); r.max_len = (int32_t) (int16_t) (fm >> 16); }
	return r;
}
template <bool UNI> J40_DEV int32_t cluster_token(DevBits &b, DevCode &c, const ClusterRegs &cl) {
	return c.use_prefix_code ? prefix_symbol<UNI>(b, c, cl.table_off, cl.fast_len, cl.max_len) : ans_symbol<UNI>(b, c, cl.table_off);
}

// LZ77 special distances, (dx + 7) * 16 + dy (spec table; cf. j40.h:2834)
#ifdef __HIPCC__
__device__
#endif
static const uint8_t LZ77_SPECIAL_DISTANCES[120] = {
	0x71, 0x80, 0x81, 0x61, 0x72, 0x90, 0x82, 0x62, 0x91, 0x51, 0x92, 0x52, 0x73, 0xa0, 0x83, 0x63, 0xa1, 0x41, 0x93, 0x53,
	0xa2, 0x42, 0x74, 0xb0, 0x84, 0x64, 0xb1, 0x31, 0xa3, 0x43, 0x94, 0x54, 0xb2, 0x32, 0x75, 0xa4, 0x44, 0xb3, 0x33, 0xc0,
	0x85, 0x65, 0xc1, 0x21, 0x95, 0x55, 0xc2, 0x22, 0xb4, 0x34, 0xa5, 0x45, 0xc3, 0x23, 0x76, 0xd0, 0x86, 0x66, 0xd1, 0x11,
	0x96, 0x56, 0xd2, 0x12, 0xb5, 0x35, 0xc4, 0x24, 0xa6, 0x46, 0xd3, 0x13, 0x77, 0xe0, 0x87, 0x67, 0xc5, 0x25, 0xe1, 0x01,
	0xb6, 0x36, 0xd4, 0x14, 0x97, 0x57, 0xe2, 0x02, 0xa7, 0x47, 0xe3, 0x03, 0xc6, 0x26, 0xd5, 0x15, 0xf0, 0xb7, 0x37, 0xe4,
	0x04, 0xf1, 0xf2, 0xd6, 0x16, 0xf3, 0xc7, 0x27, 0xe5, 0x05, 0xf4, 0xd7, 0x17, 0xe6, 0x06, 0xf5, 0xe7, 0x07, 0xf6, 0xf7,
};

// the LZ77 window holds the last `window_size` decoded integers; sections never decode more than
// window_size symbols (sized from the section's symbol bound on the host), so indices do not wrap
// before the reference's 2^20 mask would
template <bool UNI> J40_DEV int32_t code_lz77_copy(DevBits &b, DevCode &c, int32_t token, int32_t dist_mult) {
	const ClusterRegs lz = load_cluster<UNI>(c, c.num_dist - 1);
	const int32_t num_to_copy = hybrid_int_dev<UNI>(b, token - c.min_symbol, c.lz_len_cfg, c.lz_len_max_token) + c.min_length;
	token = cluster_token<UNI>(b, c, lz);
	int32_t distance = hybrid_int_dev<UNI>(b, token, lz.cfg, lz.max_token);
	if (!dist_mult) ++distance;
	else if (distance >= 120) distance -= 119;
	else {
		const int32_t special = (int32_t) uni<UNI>((uint32_t) LZ77_SPECIAL_DISTANCES[distance]);
		distance = ((special >> 4) - 7) + dist_mult * (special & 7);
		if (distance < 1) distance = 1;
	}
	if (distance > c.num_decoded) distance = c.num_decoded;
	if (distance > (1 << 20)) distance = 1 << 20;
	c.copy_pos = c.num_decoded - distance;
	c.num_to_copy = num_to_copy;
	return 0;
}

template <bool UNI> J40_DEV int32_t code_symbol(DevBits &b, DevCode &c, int32_t ctx, int32_t dist_mult, uint32_t window_size) {  // j40.h:2804
#ifdef J40_COUNT_HOOK
	J40_COUNT_HOOK;
#endif
	if (c.num_to_copy == 0) {
		const ClusterRegs cl = load_cluster<UNI>(c, ctx);
		int32_t token = cluster_token<UNI>(b, c, cl);
		if (token < c.min_symbol) {
			token = hybrid_int_dev<UNI>(b, token, cl.cfg, cl.max_token);
			if (c.lz77_enabled) {
				if (!c.window || (uint32_t) c.num_decoded >= window_size) { bits_set_error(b, ERR_TODO); return token; }
				c.window[c.num_decoded++] = token;
			}
			return token;
		}
		code_lz77_copy<UNI>(b, c, token, dist_mult);
	}
	// copy one integer out of the window
	--c.num_to_copy;
	if (!c.window || (uint32_t) c.num_decoded >= window_size) { bits_set_error(b, ERR_TODO); c.num_to_copy = 0; return 0; }
	// positions before the first decoded symbol read as zero (the reference zero-fills, j40.h:2858)
	const int32_t v = c.copy_pos < c.num_decoded ? uni<UNI>(c.window[c.copy_pos]) : 0;
	++c.copy_pos;
	c.window[c.num_decoded++] = v;
	return v;
}

template <bool UNI> J40_DEV void code_finish(DevBits &b, DevCode &c) {  // j40.h:2884
	if (!c.use_prefix_code) {
		if (c.ans_state) { if (c.ans_state != 0x130000) bits_set_error(b, ERR_ANS); }
		else { if (bits_u<UNI>(b, 16) != 0x0000) bits_set_error(b, ERR_ANS); if (bits_u<UNI>(b, 16) != 0x0013) bits_set_error(b, ERR_ANS); }
	}
}

} // namespace j40hip
