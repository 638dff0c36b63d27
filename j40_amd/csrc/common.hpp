// j40_amd/csrc/common.hpp -- error codes, integer helpers and the host-side LSB-first bit reader.
//
// Host-side product code (parses what the reference marks host-only: container, headers, TOC,
// LfGlobal / LfGroup / HfGlobal; SURVEY.md section 2 "host" rows). Written fresh; the behaviours it
// has to reproduce are cited as j40.h:line into /root/reference.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>
#include <stdexcept>

namespace j40hip {

using err_t = uint32_t;

// four-character error codes, same encoding as the reference (J40__4, j40.h:482)
constexpr err_t E4(const char (&s)[5]) {
	return ((uint32_t) (uint8_t) s[0] << 24) | ((uint32_t) (uint8_t) s[1] << 16) | ((uint32_t) (uint8_t) s[2] << 8) | (uint32_t) (uint8_t) s[3];
}

struct DecodeError { err_t code; };

[[noreturn]] inline void raise(err_t code) { throw DecodeError{code}; }
inline void should(bool cond, err_t code) { if (!cond) raise(code); }
#define J40HIP_SHOULD(cond, s) ::j40hip::should((cond), ::j40hip::E4(s))
#define J40HIP_RAISE(s) ::j40hip::raise(::j40hip::E4(s))

inline int32_t unpack_signed(int32_t x) { return (x & 1) ? -(x / 2 + 1) : x / 2; }         // j40.h:610
inline int64_t unpack_signed64(int64_t x) { return (x & 1) ? -(x / 2 + 1) : x / 2; }
inline int32_t ceil_div(int32_t x, int32_t y) { return (x + y - 1) / y; }
inline int floor_lg32(uint32_t x) { return 31 - __builtin_clz(x); }                         // x > 0
inline int ceil_lg32(uint32_t x) { return x > 1 ? 32 - __builtin_clz(x - 1) : 0; }          // x > 0 (j40.h:802)
inline int floor_lg64(uint64_t x) { return 63 - __builtin_clzll(x); }

// LSB-first bit reader over one byte range (a TOC section or the header run). Bytes are appended
// at the top of a 64-bit accumulator exactly like the reference (j40.h:1847-1882); reading past
// the range is the reference's `shrt` error.
struct BitReader {
	const uint8_t *ptr = nullptr, *end = nullptr, *begin = nullptr;
	uint64_t bits = 0;
	int nbits = 0;

	BitReader() {}
	BitReader(const uint8_t *p, size_t n) : ptr(p), end(p + n), begin(p) {}

	void refill() {
		while (nbits <= 56 && ptr < end) { bits |= (uint64_t) *ptr++ << nbits; nbits += 8; }
	}
	uint32_t u(int n) {  // n in [0, 31]
		if (nbits < n) { refill(); if (nbits < n) J40HIP_RAISE("shrt"); }
		uint32_t ret = (uint32_t) (bits & ((1ull << n) - 1));
		bits >>= n; nbits -= n;
		return ret;
	}
	uint64_t u64bits(int n) {  // n in [0, 56]
		if (nbits < n) { refill(); if (nbits < n) J40HIP_RAISE("shrt"); }
		uint64_t ret = bits & ((1ull << n) - 1);
		bits >>= n; nbits -= n;
		return ret;
	}
	// peeks up to 16 bits without failing at the end of the range (missing bits read as zero), for
	// prefix codes that may be shorter than their maximum length at the very end (j40.h:2258-2261)
	// `need` is the length the reference asks its accumulator for at this place (`nbits < max_len`, j40.h:2261): refilling at the same
	// moments keeps the accumulator's fill identical, which j40__skip's behaviour depends on (see skip_bits_like_reference)
	uint32_t peek16(int need) { if (nbits < need) refill(); return (uint32_t) (bits & 0xffff); }
	void consume(int n) {
		if (n > nbits) { bits = 0; nbits = 0; J40HIP_RAISE("shrt"); }  // j40.h:2267-2271
		bits >>= n; nbits -= n;
	}
	int32_t u32(int32_t o0, int n0, int32_t o1, int n1, int32_t o2, int n2, int32_t o3, int n3) {  // j40.h:1934
		const int32_t o[4] = {o0, o1, o2, o3}; const int n[4] = {n0, n1, n2, n3};
		uint32_t sel = u(2);
		return (int32_t) u(n[sel]) + o[sel];
	}
	int64_t u32_64(int64_t o0, int n0, int64_t o1, int n1, int64_t o2, int n2, int64_t o3, int n3) {  // j40.h:1950
		const int64_t o[4] = {o0, o1, o2, o3}; const int n[4] = {n0, n1, n2, n3};
		uint32_t sel = u(2);
		return ((int64_t) u64bits(n[sel]) + o[sel]) & (int64_t) 0xffffffff;
	}
	uint64_t u64() {  // j40.h:1966
		uint32_t sel = u(2);
		uint64_t ret = u((int) sel * 4);
		if (sel < 3) {
			ret += 17u >> (8 - sel * 4);
		} else {
			for (int shift = 12; shift < 64 && u(1); shift += 8) ret |= (uint64_t) u(shift < 56 ? 8 : 64 - shift) << shift;
		}
		return ret;
	}
	int32_t enum_() {  // j40.h:1979
		int32_t v = u32(0, 0, 1, 0, 2, 4, 18, 6);
		J40HIP_SHOULD(v < 31, "enum");
		return v;
	}
	float f16();  // j40.h:1987
	int32_t u8() {  // j40.h:1994
		if (u(1)) { int n = (int) u(3); return (int32_t) u(n) + (1 << n); }
		return 0;
	}
	int32_t at_most(int32_t max) {  // j40.h:2004
		int32_t v = max > 0 ? (int32_t) u(ceil_lg32((uint32_t) max + 1)) : 0;
		J40HIP_SHOULD(v <= max, "rnge");
		return v;
	}
	void zero_pad_to_byte() {  // j40.h:1884
		int n = nbits & 7;
		J40HIP_SHOULD((bits & ((1u << n) - 1)) == 0, "pad0");
		bits >>= n; nbits -= n;
	}
	// j40__skip as the reference has it (j40.h:1892-1911): when the accumulator already holds n bits it drops them and then falls
	// through to the general case with n unchanged, i.e. it skips another n >> 3 bytes and n & 7 bits -- 2n bits in all.
	// Header fields that are "skipped" (extensions, the EPF sigma placeholder) land where the reference lands only this way;
	// the accumulator is filled exactly like the reference's, so the condition is the same.
	void skip_bits_like_reference(int64_t n) {
		if (nbits >= n) { bits >>= (int) n; nbits -= (int) n; }
		else { n -= nbits; bits = 0; nbits = 0; }
		int64_t bytes = n >> 3;
		J40HIP_SHOULD(end - ptr >= bytes, "shrt");
		ptr += bytes;
		(void) u((int) (n & 7));
	}
	void skip_bits(int64_t n) {  // what j40__skip means to do
		if (nbits >= n) { bits >>= (int) n; nbits -= (int) n; return; }
		n -= nbits; bits = 0; nbits = 0;
		int64_t bytes = n >> 3;
		J40HIP_SHOULD(end - ptr >= bytes, "shrt");
		ptr += bytes;
		(void) u((int) (n & 7));
	}
	// position of the next unread bit, relative to the start of the range
	size_t bit_position() const { return (size_t) (ptr - begin) * 8 - (size_t) nbits; }
	// byte position after zero_pad_to_byte
	size_t byte_position() const { return (size_t) (ptr - begin) - (size_t) (nbits >> 3); }
	void no_more_bytes() {  // j40.h:2011: the section must end exactly here
		zero_pad_to_byte();
		J40HIP_SHOULD(nbits == 0 && ptr == end, "excs");
	}
};

} // namespace j40hip
