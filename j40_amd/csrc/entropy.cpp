// j40_amd/csrc/entropy.cpp -- see entropy.hpp
#include "entropy.hpp"
#include <cmath>
#include <algorithm>

namespace j40hip {

float BitReader::f16() {  // j40.h:1987
	int32_t b = (int32_t) u(16);
	int32_t biased_exp = (b >> 10) & 0x1f;
	J40HIP_SHOULD(biased_exp != 31, "!fin");
	return (float) ((b >> 15) ? -1 : 1) * ldexpf((float) ((b & 0x3ff) | (biased_exp > 0 ? 0x400 : 0)), biased_exp - 25);
}

HybridCfg read_hybrid_cfg(BitReader &br, int32_t log_alpha_size) {  // j40.h:2297
	HybridCfg c;
	c.split_exp = (int8_t) br.at_most(log_alpha_size);
	if (c.split_exp != log_alpha_size) {
		c.msb_in_token = (int8_t) br.at_most(c.split_exp);
		c.lsb_in_token = (int8_t) br.at_most(c.split_exp - c.msb_in_token);
	}
	c.max_token = (1 << c.split_exp) + ((30 - c.split_exp) << (c.lsb_in_token + c.msb_in_token)) - 1;
	return c;
}

static inline int32_t hybrid_int(BitReader &br, int32_t token, const HybridCfg &c) {  // j40.h:2313
	int32_t split = 1 << c.split_exp;
	if (token < split) return token;
	J40HIP_SHOULD(token <= c.max_token, "iovf");
	int32_t in_token = c.msb_in_token + c.lsb_in_token;
	int32_t midbits = c.split_exp - in_token + ((token - split) >> in_token);
	int32_t mid = (int32_t) br.u(midbits);
	int32_t top = 1 << c.msb_in_token;
	int32_t lo = token & ((1 << c.lsb_in_token) - 1);
	int32_t hi = (token >> c.lsb_in_token) & (top - 1);
	return ((top | hi) << (midbits + c.lsb_in_token)) | ((mid << c.lsb_in_token) | lo);
}

// ------------------------------------------------------------------------------------------------
// prefix codes (RFC 7932 section 3)

static inline int32_t prefix_decode(BitReader &br, int32_t fast_len, int32_t max_len, const int32_t *table, int32_t need = 0) {  // j40.h:2256
	uint32_t window = br.peek16(need ? need : max_len);
	int32_t entry = table[window & ((1u << fast_len) - 1)];
	int32_t used = 0;
	if (entry < 0 && fast_len < max_len) {
		const int32_t *ovf = table - entry;
		uint32_t rest = window >> fast_len;
		int32_t code_len;
		do { entry = *ovf++; code_len = entry & 15; } while ((uint32_t) ((entry >> 4) & 0xfff) != (rest & ((1u << code_len) - 1)));
		used = fast_len;
	}
	br.consume(used + (entry & 15));
	return entry >> 16;
}

static uint32_t reverse_bits(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((v >> i) & 1) << (n - 1 - i); return r; }

// Builds the lookup table for a canonical code given per-symbol lengths (<= 15). Codes are
// consumed LSB-first from the bit buffer, so LUT indices are bit-reversed canonical codes.
// Same table format as the reference (j40.h:2030-2043) because the HIP kernels read it too.
static void build_prefix_table(const std::vector<int32_t> &lengths, Cluster *out) {
	const int32_t MAXLEN = 15, TYPICAL_FAST = 7, GROWTH = 2;
	int32_t counts[MAXLEN + 1] = {0};
	for (int32_t l : lengths) ++counts[l];
	counts[0] = 0;
	int32_t max_len = 1;
	for (int32_t l = 1; l <= MAXLEN; ++l) if (counts[l]) max_len = l;
	int32_t fast_len;
	if (max_len <= TYPICAL_FAST) fast_len = max_len;
	else {  // same sizing rule as j40.h:2190-2206 so that table sizes (and LDS budgets) agree
		int32_t size = 1 << TYPICAL_FAST;
		fast_len = TYPICAL_FAST;
		for (int32_t l = fast_len + 1; l <= max_len; ++l) size += counts[l];
		int32_t limit = size * GROWTH;
		for (int32_t l = TYPICAL_FAST + 1; l <= max_len; ++l) {
			size = size + (1 << l) - counts[l];
			if (size <= limit) fast_len = l;
		}
	}
	int32_t novf = 0;
	for (int32_t l = fast_len + 1; l <= max_len; ++l) novf += counts[l];
	std::vector<int32_t> table((size_t) (1 << fast_len) + (size_t) novf + 1, 0);
	// canonical codes, shortest first then by symbol
	uint32_t next_code[MAXLEN + 2] = {0};
	{ uint32_t c = 0; for (int32_t l = 1; l <= MAXLEN; ++l) { c = (c + (uint32_t) counts[l - 1]) << 1; next_code[l] = c; } }
	// overflow entries are grouped by their fast_len-bit prefix; within a group any order works as
	// long as exactly one entry matches. Place groups in order of first appearance.
	struct Ovf { uint32_t prefix; int32_t entry; };
	std::vector<Ovf> ovf;
	for (size_t s = 0; s < lengths.size(); ++s) {
		int32_t l = lengths[s];
		if (!l) continue;
		uint32_t code = next_code[l]++;
		uint32_t rev = reverse_bits(code, l);  // bit i = i-th bit read
		if (l <= fast_len) {
			for (uint32_t idx = rev; idx < (1u << fast_len); idx += 1u << l) table[idx] = (int32_t) ((uint32_t) s << 16) | l;
		} else {
			ovf.push_back({rev & ((1u << fast_len) - 1), (int32_t) (((uint32_t) s << 16) | ((rev >> fast_len) << 4) | (uint32_t) (l - fast_len))});
		}
	}
	int32_t pos = 1 << fast_len;
	std::vector<bool> done(ovf.size(), false);
	for (size_t i = 0; i < ovf.size(); ++i) {
		if (done[i]) continue;
		table[ovf[i].prefix] = -pos;
		for (size_t j = i; j < ovf.size(); ++j) if (!done[j] && ovf[j].prefix == ovf[i].prefix) { table[(size_t) pos++] = ovf[j].entry; done[j] = true; }
	}
	out->fast_len = fast_len; out->max_len = max_len; out->table.swap(table);
	out->lengths.assign(lengths.begin(), lengths.end());
}

bool rfc_simple_code_order() {
	static const bool on = [] { const char *e = getenv("J40HIP_RFC_SIMPLE_CODES"); return e && atoi(e) != 0; }();
	return on;
}

static void read_prefix_tree(BitReader &br, int32_t alphabet, Cluster *out) {  // j40.h:2049
	if (alphabet == 1) { out->fast_len = out->max_len = 0; out->table.assign(1, 0); out->lengths.assign(1, 0); return; }
	int32_t hskip = (int32_t) br.u(2);
	if (hskip == 1) {  // simple code, RFC 7932 section 3.4
		int32_t nsym = (int32_t) br.u(2) + 1, syms[4] = {0, 0, 0, 0};
		for (int32_t i = 0; i < nsym; ++i) {
			syms[i] = br.at_most(alphabet - 1);
			for (int32_t j = 0; j < i; ++j) J40HIP_SHOULD(syms[i] != syms[j], "hufd");
		}
		std::vector<int32_t> lengths((size_t) alphabet, 0);
		bool tree_select = nsym == 4 && br.u(1);
		switch (nsym) {
		case 1: out->fast_len = out->max_len = 0; out->table.assign(1, syms[0] << 16); out->lengths.assign((size_t) alphabet, 0); out->lengths[(size_t) syms[0]] = 255; return;  // 255: the only symbol, zero bits
		case 2: lengths[(size_t) syms[0]] = lengths[(size_t) syms[1]] = 1; break;
		case 3: lengths[(size_t) syms[0]] = 1; lengths[(size_t) syms[1]] = lengths[(size_t) syms[2]] = 2; break;
		default:
			if (tree_select) { lengths[(size_t) syms[0]] = 1; lengths[(size_t) syms[1]] = 2; lengths[(size_t) syms[2]] = lengths[(size_t) syms[3]] = 3; }
			else if (rfc_simple_code_order()) for (int i = 0; i < 4; ++i) lengths[(size_t) syms[i]] = 2;   // RFC 7932: canonical code over the sorted symbols
			else {
				// The reference fills this template's table with symref {0, 1, 2, 3} at the index made of the bits in READ order
				// (j40.h:2090, 2112): the sorted symbols 1 and 2 swap places against RFC 7932's canonical code ("10" reads as index 1).
				// Same table here by default -- results identical to the reference's; J40HIP_RFC_SIMPLE_CODES=1 selects the RFC order.
				// The view marks the four symbols with length 2 | 128 so that a table can be rebuilt from the lengths alone.
				std::sort(syms, syms + 4);
				out->fast_len = out->max_len = 2;
				out->table.assign(5, 0);
				for (int i = 0; i < 4; ++i) out->table[(size_t) i] = (syms[i] << 16) | 2;
				out->lengths.assign((size_t) alphabet, 0);
				for (int i = 0; i < 4; ++i) out->lengths[(size_t) syms[i]] = 2 | 128;
				return;
			}
		}
		build_prefix_table(lengths, out);
		return;
	}
	// complex code, RFC 7932 section 3.5: code-length code first
	static const uint8_t ORDER[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
	std::vector<int32_t> l1len(18, 0);
	int32_t total = 0, nread = hskip, nzero = hskip;
	for (; nread < 18 && total < 32; ++nread) {
		// fixed code over 0..5: 00->0 01->3 10->4 110->2 1110->1 1111->5 (bits in read order)
		int32_t v;
		uint32_t w = br.peek16(4);   // L0MAXLEN (j40.h:2120)
		if ((w & 3) == 0) { v = 0; br.consume(2); }
		else if ((w & 3) == 2) { v = 3; br.consume(2); }
		else if ((w & 3) == 1) { v = 4; br.consume(2); }
		else if ((w & 7) == 3) { v = 2; br.consume(3); }
		else if ((w & 15) == 7) { v = 1; br.consume(4); }
		else { v = 5; br.consume(4); }
		l1len[ORDER[nread]] = v;
		if (v) total += 32 >> v; else ++nzero;
	}
	J40HIP_SHOULD(total == 32 && nzero != nread, "hufd");
	Cluster l1;
	build_prefix_table(l1len, &l1);
	std::vector<int32_t> lengths((size_t) alphabet, 0);
	int32_t prev = 8, rep_nonzero = 0, rep_zero = 0, i = 0;
	total = 0;
	while (i < alphabet && total < 32768) {
		int32_t code = prefix_decode(br, l1.fast_len, l1.max_len, l1.table.data(), 5);   // L1MAXLEN (j40.h:2150)
		if (code < 16) {
			lengths[(size_t) i++] = code;
			if (code) { total += 32768 >> code; prev = code; }
			rep_nonzero = rep_zero = 0;
		} else if (code == 16) {  // repeat previous non-zero length; consecutive 16s extend the run
			rep_zero = 0;
			int32_t old = rep_nonzero;
			rep_nonzero = (old > 0 ? 4 * (old - 2) : 0) + 3 + (int32_t) br.u(2);
			int32_t extra = rep_nonzero - old;
			J40HIP_SHOULD(i + extra <= alphabet, "hufd");
			for (int32_t k = 0; k < extra; ++k) lengths[(size_t) i++] = prev;
			total += (32768 >> prev) * extra;
		} else {  // 17: repeat zero
			rep_nonzero = 0;
			int32_t old = rep_zero;
			rep_zero = (old > 0 ? 8 * (old - 2) : 0) + 3 + (int32_t) br.u(3);
			int32_t extra = rep_zero - old;
			J40HIP_SHOULD(i + extra <= alphabet, "hufd");
			i += extra;
		}
	}
	J40HIP_SHOULD(total == 32768, "hufd");
	build_prefix_table(lengths, out);
}

// ------------------------------------------------------------------------------------------------
// rANS distributions and alias tables

static void read_ans_distribution(BitReader &br, int32_t log_alpha_size, std::vector<int16_t> *outD) {  // j40.h:2601
	const int32_t table_size = 1 << log_alpha_size, DISTSUM = 4096;
	std::vector<int16_t> D((size_t) table_size, 0);
	switch (br.u(2)) {
	case 1: {
		int32_t v = br.u8();
		J40HIP_SHOULD(v < table_size, "ansd");
		D[(size_t) v] = (int16_t) DISTSUM;
		break;
	}
	case 3: {
		int32_t v1 = br.u8(), v2 = br.u8();
		J40HIP_SHOULD(v1 != v2 && v1 < table_size && v2 < table_size, "ansd");
		D[(size_t) v1] = (int16_t) br.u(12);
		D[(size_t) v2] = (int16_t) (DISTSUM - D[(size_t) v1]);
		break;
	}
	case 2: {
		int32_t alpha = br.u8() + 1;
		J40HIP_SHOULD(alpha <= table_size, "ansd");
		int32_t d = DISTSUM / alpha, bias = DISTSUM % alpha;
		for (int32_t i = 0; i < alpha; ++i) D[(size_t) i] = (int16_t) (d + (i < bias));
		break;
	}
	default: {
		int32_t len = br.u(1) ? (br.u(1) ? (br.u(1) ? 3 : 2) : 1) : 0;
		int32_t shift = (int32_t) br.u(len) + (1 << len) - 1;
		J40HIP_SHOULD(shift <= 13, "ansd");
		int32_t alpha = br.u8() + 3;
		// log-count code (kLogCountLut): 4-bit primary table, values >= 11 via the `1000` escape
		static const int8_t PRIMARY_VAL[16] = {10, -1, 7, 3, 6, 8, 9, 5, 10, 4, 7, 1, 6, 8, 9, 2};
		static const int8_t PRIMARY_LEN[16] = {3, 0, 3, 4, 3, 3, 3, 4, 3, 4, 3, 4, 3, 3, 3, 4};
		std::vector<int32_t> codes;  // >= 0 log count, < 0 negated repeat
		int32_t omit_log = -1, i = 0;
		while (i < alpha) {
			uint32_t w = br.peek16(7);   // j40.h:2655
			int32_t code;
			if ((w & 15) != 1) { code = PRIMARY_VAL[w & 15]; br.consume(PRIMARY_LEN[w & 15]); }
			else if ((w >> 4) & 1) { code = 0; br.consume(5); }
			else if (((w >> 4) & 3) == 2) { code = 11; br.consume(6); }
			else if (((w >> 4) & 7) == 0) { code = 12; br.consume(7); }
			else { code = 13; br.consume(7); }
			if (code < 13) { ++i; codes.push_back(code); if (omit_log < code) omit_log = code; }
			else { int32_t rep = br.u8() + 4; i += rep; codes.push_back(-rep); }
		}
		J40HIP_SHOULD(i == alpha && omit_log >= 0, "ansd");
		int32_t omit_pos = -1, n = 0, total = 0;
		for (size_t k = 0; k < codes.size() && n < table_size; ++k) {
			int32_t code = codes[k];
			if (code < 0) {
				int16_t prevd = n > 0 ? D[(size_t) n - 1] : 0;
				J40HIP_SHOULD(prevd >= 0, "ansd");
				int32_t rep = -code < table_size - n ? -code : table_size - n;
				total += (int32_t) prevd * rep;
				while (rep-- > 0) D[(size_t) n++] = prevd;
			} else if (code == omit_log) {
				omit_pos = n; omit_log = -1; D[(size_t) n++] = -1;
			} else if (code < 2) {
				total += code; D[(size_t) n++] = (int16_t) code;
			} else {
				--code;
				int32_t bitcount = shift - ((12 - code) >> 1);
				if (bitcount < 0) bitcount = 0;
				if (bitcount > code) bitcount = code;
				int32_t v = (1 << code) + ((int32_t) br.u(bitcount) << (code - bitcount));
				total += v; D[(size_t) n++] = (int16_t) v;
			}
		}
		J40HIP_SHOULD(omit_pos >= 0 && total <= DISTSUM, "ansd");
		D[(size_t) omit_pos] = (int16_t) (DISTSUM - total);
	} }
	outD->swap(D);
}

// alias table construction as specified by the format (j40__init_alias_map, j40.h:2362): the
// resulting mapping is normative because the encoder's symbol slots depend on it
static void build_alias_table(const std::vector<int16_t> &D, int32_t log_alpha_size, std::vector<AnsEntry> *out) {
	const int32_t table_size = 1 << log_alpha_size, log_bucket = 12 - log_alpha_size, bucket_size = 1 << log_bucket;
	std::vector<int32_t> cutoff((size_t) table_size, 0), link((size_t) table_size, 0), symbol((size_t) table_size, 0);
	int32_t first = 0, second;
	while (first < table_size && !D[(size_t) first]) ++first;
	second = first + 1;
	while (second < table_size && !D[(size_t) second]) ++second;
	if (first < table_size && second >= table_size) {  // a single symbol owns everything
		for (int32_t j = 0; j < table_size; ++j) { symbol[(size_t) j] = first; link[(size_t) j] = j << log_bucket; cutoff[(size_t) j] = 0; }
	} else {
		int32_t under = -1, over = -1;  // intrusive stacks threaded through `link`
		for (int32_t i = 0; i < table_size; ++i) {
			cutoff[(size_t) i] = D[(size_t) i];
			if (cutoff[(size_t) i] > bucket_size) { link[(size_t) i] = over; over = i; }
			else if (cutoff[(size_t) i] < bucket_size) { link[(size_t) i] = under; under = i; }
			else { symbol[(size_t) i] = i; link[(size_t) i] = 0; }
		}
		while (over >= 0) {
			J40HIP_SHOULD(under >= 0, "ansd");
			int32_t u = under, o = over;
			int32_t by = bucket_size - cutoff[(size_t) u];
			under = link[(size_t) u];
			cutoff[(size_t) o] -= by;
			symbol[(size_t) u] = o;
			link[(size_t) u] = cutoff[(size_t) o] - cutoff[(size_t) u];  // becomes the offset
			if (cutoff[(size_t) o] < bucket_size) { over = link[(size_t) o]; link[(size_t) o] = under; under = o; }
			else if (cutoff[(size_t) o] == bucket_size) { over = link[(size_t) o]; link[(size_t) o] = 0; symbol[(size_t) o] = o; }
		}
	}
	out->assign((size_t) table_size, 0);
	for (int32_t i = 0; i < table_size; ++i) {
		uint64_t e = (uint64_t) (uint32_t) cutoff[(size_t) i] | ((uint64_t) (uint32_t) (link[(size_t) i] & 0xfff) << 8) |
			((uint64_t) (uint32_t) symbol[(size_t) i] << 20) | ((uint64_t) (uint32_t) D[(size_t) symbol[(size_t) i]] << 28) |
			((uint64_t) (uint32_t) (D[(size_t) i] < 0 ? 0 : D[(size_t) i]) << 41);
		(*out)[(size_t) i] = e;
	}
}

static inline int32_t ans_decode(BitReader &br, uint32_t *state, int32_t log_bucket, const AnsEntry *alias) {  // j40.h:2441
	if (*state == 0) { *state = br.u(16); *state |= br.u(16) << 16; }
	uint32_t idx = *state & 0xfff, i = idx >> log_bucket, pos = idx & ((1u << log_bucket) - 1);
	AnsEntry e = alias[i];
	bool aliased = pos >= (uint32_t) (e & 0xff);
	uint32_t symbol = aliased ? (uint32_t) (e >> 20) & 0xff : i;
	uint32_t offset = aliased ? (uint32_t) (e >> 8) & 0xfff : 0;
	uint32_t d = aliased ? (uint32_t) (e >> 28) & 0x1fff : (uint32_t) (e >> 41) & 0x1fff;
	*state = d * (*state >> 12) + offset + pos;
	if (*state < (1u << 16)) *state = (*state << 16) | br.u(16);
	return (int32_t) symbol;
}

// ------------------------------------------------------------------------------------------------

void read_cluster_map(BitReader &br, int32_t num_dist, int32_t max_allowed, int32_t *num_clusters, std::vector<uint8_t> *outmap) {  // j40.h:2526
	if (max_allowed > num_dist) max_allowed = num_dist;
	std::vector<uint8_t> map((size_t) num_dist, 0);
	if (num_dist == 1) { *num_clusters = 1; outmap->swap(map); return; }
	if (br.u(1)) {  // simple: fixed-width entries
		int32_t nbits = (int32_t) br.u(2);
		for (int32_t i = 0; i < num_dist; ++i) {
			map[(size_t) i] = (uint8_t) br.u(nbits);
			J40HIP_SHOULD((int32_t) map[(size_t) i] < max_allowed, "clst");
		}
	} else {
		bool use_mtf = br.u(1);
		CodeSpec nested;
		read_code_spec(br, num_dist <= 2 ? -1 : 1, &nested);
		CodeState code(&nested);
		for (int32_t i = 0; i < num_dist; ++i) {
			int32_t index = decode_symbol(br, code, 0, 0);
			J40HIP_SHOULD(index < max_allowed, "clst");
			map[(size_t) i] = (uint8_t) index;
		}
		finish_code(br, code);
		if (use_mtf) {
			uint8_t mtf[256];
			for (int i = 0; i < 256; ++i) mtf[i] = (uint8_t) i;
			for (int32_t i = 0; i < num_dist; ++i) {
				int j = map[(size_t) i];
				uint8_t moved = mtf[j];
				map[(size_t) i] = moved;
				for (; j > 0; --j) mtf[j] = mtf[j - 1];
				mtf[0] = moved;
			}
		}
	}
	// cluster ids must be exactly 0..n-1 (j40.h:2584-2588)
	bool seen[256] = {false};
	for (uint8_t m : map) seen[m] = true;
	int32_t n = 0;
	while (n < 256 && seen[n]) ++n;
	for (int32_t i = n; i < 256; ++i) J40HIP_SHOULD(!seen[i], "clst");
	*num_clusters = n;
	outmap->swap(map);
}

void read_code_spec(BitReader &br, int32_t num_dist, CodeSpec *spec) {  // j40.h:2711
	bool allow_lz77 = num_dist > 0;
	if (num_dist < 0) num_dist = -num_dist;
	*spec = CodeSpec();
	spec->lz77_enabled = br.u(1);
	if (spec->lz77_enabled) {
		J40HIP_SHOULD(allow_lz77, "lz77");
		spec->min_symbol = br.u32(224, 0, 512, 0, 4096, 0, 8, 15);
		spec->min_length = br.u32(3, 0, 4, 0, 5, 2, 9, 8);
		spec->lz_len_cfg = read_hybrid_cfg(br, 8);
		++num_dist;  // the last context is the LZ77 distance context
	}
	read_cluster_map(br, num_dist, 256, &spec->num_clusters, &spec->cluster_map);
	spec->clusters.assign((size_t) spec->num_clusters, Cluster());
	spec->use_prefix_code = br.u(1);
	if (spec->use_prefix_code) {
		for (auto &c : spec->clusters) c.cfg = read_hybrid_cfg(br, 15);
		std::vector<int32_t> counts((size_t) spec->num_clusters, 1);
		for (auto &cnt : counts) {
			if (br.u(1)) {
				int32_t n = (int32_t) br.u(4);
				cnt = 1 + (1 << n) + (int32_t) br.u(n);
				J40HIP_SHOULD(cnt <= (1 << 15), "hufd");
			}
		}
		for (int32_t i = 0; i < spec->num_clusters; ++i) read_prefix_tree(br, counts[(size_t) i], &spec->clusters[(size_t) i]);
	} else {
		spec->log_alpha_size = 5 + (int32_t) br.u(2);
		for (auto &c : spec->clusters) c.cfg = read_hybrid_cfg(br, spec->log_alpha_size);
		for (auto &c : spec->clusters) {
			read_ans_distribution(br, spec->log_alpha_size, &c.D);
			build_alias_table(c.D, spec->log_alpha_size, &c.alias);
		}
	}
	spec->num_dist = num_dist;
}

void finish_code_spec_tables(CodeSpec *spec) {
	auto max_token = [](HybridCfg &c) { c.max_token = (1 << c.split_exp) + ((30 - c.split_exp) << (c.lsb_in_token + c.msb_in_token)) - 1; };   // as read_hybrid_cfg
	max_token(spec->lz_len_cfg);
	for (Cluster &cl : spec->clusters) {
		max_token(cl.cfg);
		if (spec->use_prefix_code) {
			// the two zero-bit forms first (read_prefix_tree): an alphabet of one symbol, a simple code naming a single symbol (255)
			int32_t only = -1;
			for (size_t i = 0; i < cl.lengths.size(); ++i) if (cl.lengths[i] == 255) only = (int32_t) i;
			if (cl.lengths.size() <= 1) { cl.fast_len = cl.max_len = 0; cl.table.assign(1, 0); cl.lengths.assign(1, 0); }
			else if (only >= 0) { cl.fast_len = cl.max_len = 0; cl.table.assign(1, only << 16); }
			else {
				std::vector<int32_t> lengths(cl.lengths.begin(), cl.lengths.end());
				{   // the reference-ordered NSYM = 4 template (read_prefix_tree): four symbols marked 2 | 128
					std::vector<int32_t> marked;
					for (size_t i = 0; i < lengths.size(); ++i) if (lengths[i] == (2 | 128)) marked.push_back((int32_t) i);
					if (marked.size() == 4) {
						cl.fast_len = cl.max_len = 2; cl.table.assign(5, 0);
						for (int i = 0; i < 4; ++i) cl.table[(size_t) i] = (marked[(size_t) i] << 16) | 2;
						continue;
					}
				}
				for (int32_t l : lengths) if (l > 15) J40HIP_RAISE("hufd");
				build_prefix_table(lengths, &cl);
			}
		} else {
			if ((int32_t) cl.D.size() != (1 << spec->log_alpha_size)) J40HIP_RAISE("ans?");
			build_alias_table(cl.D, spec->log_alpha_size, &cl.alias);
		}
	}
}

static inline int32_t cluster_token(BitReader &br, const CodeSpec *spec, const Cluster &cl, uint32_t *ans_state) {
	if (spec->use_prefix_code) return prefix_decode(br, cl.fast_len, cl.max_len, cl.table.data());
	return ans_decode(br, ans_state, 12 - spec->log_alpha_size, cl.alias.data());
}

int32_t decode_symbol(BitReader &br, CodeState &code, int32_t ctx, int32_t dist_mult) {  // j40.h:2804
	static const int32_t MASK = 0xfffff;
	const CodeSpec *spec = code.spec;
	if (code.num_to_copy > 0) {
		--code.num_to_copy;
		int32_t v = code.window[(size_t) (code.copy_pos++ & MASK)];
		code.window[(size_t) (code.num_decoded++ & MASK)] = v;
		return v;
	}
	const Cluster &cl = spec->clusters[spec->cluster_map[(size_t) ctx]];
	int32_t token = cluster_token(br, spec, cl, &code.ans_state);
	if (token >= spec->min_symbol) {  // LZ77 copy
		const Cluster &lz = spec->clusters[spec->cluster_map[(size_t) spec->num_dist - 1]];
		int32_t num_to_copy = hybrid_int(br, token - spec->min_symbol, spec->lz_len_cfg) + spec->min_length;
		token = cluster_token(br, spec, lz, &code.ans_state);
		int32_t distance = hybrid_int(br, token, lz.cfg);
		if (!dist_mult) {
			++distance;
		} else if (distance >= 120) {
			distance -= 119;
		} else {
			// special distances, the spec's table of (dx, dy) pairs packed as (dx + 7) * 16 + dy
			static const uint8_t SPECIAL[120] = {
				0x71, 0x80, 0x81, 0x61, 0x72, 0x90, 0x82, 0x62, 0x91, 0x51, 0x92, 0x52, 0x73, 0xa0, 0x83, 0x63, 0xa1, 0x41, 0x93, 0x53,
				0xa2, 0x42, 0x74, 0xb0, 0x84, 0x64, 0xb1, 0x31, 0xa3, 0x43, 0x94, 0x54, 0xb2, 0x32, 0x75, 0xa4, 0x44, 0xb3, 0x33, 0xc0,
				0x85, 0x65, 0xc1, 0x21, 0x95, 0x55, 0xc2, 0x22, 0xb4, 0x34, 0xa5, 0x45, 0xc3, 0x23, 0x76, 0xd0, 0x86, 0x66, 0xd1, 0x11,
				0x96, 0x56, 0xd2, 0x12, 0xb5, 0x35, 0xc4, 0x24, 0xa6, 0x46, 0xd3, 0x13, 0x77, 0xe0, 0x87, 0x67, 0xc5, 0x25, 0xe1, 0x01,
				0xb6, 0x36, 0xd4, 0x14, 0x97, 0x57, 0xe2, 0x02, 0xa7, 0x47, 0xe3, 0x03, 0xc6, 0x26, 0xd5, 0x15, 0xf0, 0xb7, 0x37, 0xe4,
				0x04, 0xf1, 0xf2, 0xd6, 0x16, 0xf3, 0xc7, 0x27, 0xe5, 0x05, 0xf4, 0xd7, 0x17, 0xe6, 0x06, 0xf5, 0xe7, 0x07, 0xf6, 0xf7,
			};
			int32_t special = SPECIAL[distance];
			distance = ((special >> 4) - 7) + dist_mult * (special & 7);
			if (distance < 1) distance = 1;
		}
		if (distance > code.num_decoded) distance = code.num_decoded;
		if (distance > (1 << 20)) distance = 1 << 20;
		code.copy_pos = code.num_decoded - distance;
		if (code.window.empty()) code.window.assign((size_t) 1 << 20, 0);  // zero filled like the reference's calloc (j40.h:2858)
		code.num_to_copy = num_to_copy - 1;
		int32_t v = code.window[(size_t) (code.copy_pos++ & MASK)];
		code.window[(size_t) (code.num_decoded++ & MASK)] = v;
		return v;
	}
	token = hybrid_int(br, token, cl.cfg);
	if (spec->lz77_enabled) {
		if (code.window.empty()) code.window.assign((size_t) 1 << 20, 0);
		code.window[(size_t) (code.num_decoded++ & MASK)] = token;
	}
	return token;
}

void finish_code(BitReader &br, CodeState &code) {  // j40.h:2884
	if (!code.spec->use_prefix_code) {
		if (code.ans_state) {
			J40HIP_SHOULD(code.ans_state == 0x130000, "ans?");
		} else {
			J40HIP_SHOULD(br.u(16) == 0x0000, "ans?");
			J40HIP_SHOULD(br.u(16) == 0x0013, "ans?");
		}
	}
	code.window.clear(); code.window.shrink_to_fit();
}

std::vector<int32_t> read_permutation(BitReader &br, CodeState &code, int32_t size, int32_t skip) {  // j40.h:5428
	int32_t end = decode_symbol(br, code, ceil_lg32((uint32_t) size + 1) < 7 ? ceil_lg32((uint32_t) size + 1) : 7, 0);
	J40HIP_SHOULD(end <= size - skip, "perm");
	std::vector<int32_t> out((size_t) end);
	int32_t prev = 0;
	for (int32_t i = 0; i < end; ++i) {
		int32_t c = ceil_lg32((uint32_t) prev + 1);
		prev = out[(size_t) i] = decode_symbol(br, code, c < 7 ? c : 7, 0);
		J40HIP_SHOULD(prev < size - (skip + i), "perm");
	}
	return out;
}

} // namespace j40hip
