// tools/jxlsynth_common.hpp -- bit writer + entropy *encoder* side used by the synthetic stream
// generator (test/bench infrastructure; nothing here is linked into the product library).
//
// There is no JPEG XL encoder and no .jxl file in this environment, so every
// test/bench input is produced by this writer. It emits exactly the syntax the reference reader
// consumes; each routine cites the reader it is the inverse of (file:line into /root/reference).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <array>
#include <algorithm>
#include <string>

namespace synth {

[[noreturn]] inline void die(const char *msg) { fprintf(stderr, "jxlsynth: %s\n", msg); exit(2); }

inline int floor_lg(uint32_t x) { return 31 - __builtin_clz(x); }            // x > 0
inline int ceil_lg(uint32_t x) { return x > 1 ? 32 - __builtin_clz(x - 1) : 0; }  // x > 0
inline uint32_t pack_signed(int32_t v) { return v >= 0 ? (uint32_t) v * 2 : (uint32_t) (-(int64_t) v) * 2 - 1; }  // inverse of j40.h:610

struct SplitMix64 {
	uint64_t s;
	explicit SplitMix64(uint64_t seed) : s(seed) {}
	uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
	uint32_t below(uint32_t n) { return (uint32_t) ((next() >> 11) % n); }
	double unit() { return (double) (next() >> 11) * (1.0 / 9007199254740992.0); }
};

// LSB-first bit writer: inverse of j40__u (j40.h:1914) / j40__always_refill (j40.h:1847)
struct BitWriter {
	std::vector<uint8_t> bytes;
	uint64_t acc = 0;
	int nacc = 0;
	void put(uint64_t v, int n) {
		while (n > 0) {
			int take = std::min(n, 32);
			acc |= (v & ((1ull << take) - 1)) << nacc;
			nacc += take;
			v >>= take; n -= take;
			while (nacc >= 8) { bytes.push_back((uint8_t) acc); acc >>= 8; nacc -= 8; }
		}
	}
	void pad() { if (nacc) { bytes.push_back((uint8_t) acc); acc = 0; nacc = 0; } }  // zero bits up to the byte boundary (j40.h:1884)
	size_t bitpos() const { return bytes.size() * 8 + (size_t) nacc; }
	// U32(o0,n0,...): 2-bit selector then n[sel] bits of (v - o[sel]) (j40.h:1934)
	void u32(int64_t v, int64_t o0, int n0, int64_t o1, int n1, int64_t o2, int n2, int64_t o3, int n3) {
		const int64_t o[4] = {o0, o1, o2, o3}; const int n[4] = {n0, n1, n2, n3};
		for (int s = 0; s < 4; ++s) if (v >= o[s] && v - o[s] < ((int64_t) 1 << n[s])) { put((uint64_t) s, 2); put((uint64_t) (v - o[s]), n[s]); return; }
		die("u32: value not representable");
	}
	void at_most(int v, int max) { if (max > 0) put((uint64_t) v, ceil_lg((uint32_t) max + 1)); }  // j40.h:2004
	void u8(int v) {  // j40.h:1994
		if (v == 0) { put(0, 1); return; }
		int n = floor_lg((uint32_t) v);
		put(1, 1); put((uint64_t) n, 3); put((uint64_t) (v - (1 << n)), n);
	}
	void u64(uint64_t v) {  // j40.h:1966 (only the small forms are needed)
		if (v == 0) { put(0, 2); return; }
		if (v <= 16) { put(1, 2); put(v - 1, 4); return; }
		if (v <= 272) { put(2, 2); put(v - 17, 8); return; }
		put(3, 2); put(v & 0xfff, 12); v >>= 12;
		int shift = 12;
		while (v && shift < 60) { put(1, 1); put(v & 0xff, 8); v >>= 8; shift += 8; }
		if (shift < 60) put(0, 1); else if (v) { put(1, 1); put(v & 0xf, 4); } else put(0, 1);
	}
	void f16(float f) {  // inverse of j40.h:1987 for values that are exactly representable
		if (f == 0.0f) { put(0, 16); return; }
		int sign = f < 0; float a = sign ? -f : f; int e = 0;
		while (a >= 2048.0f) { a *= 0.5f; ++e; }
		while (a < 1024.0f) { a *= 2.0f; --e; }
		int biased = e + 25; int mant = (int) a;
		if ((float) mant != a) die("f16: value not exactly representable");
		if (biased <= 0) { while (biased <= 0) { mant >>= 1; ++biased; } put((uint64_t) ((sign << 15) | (mant & 0x3ff)), 16); return; }
		if (biased >= 31) die("f16: too large");
		put((uint64_t) ((sign << 15) | (biased << 10) | (mant & 0x3ff)), 16);
	}
	void append_bytes(const std::vector<uint8_t> &b) { if (nacc) die("append on unaligned writer"); bytes.insert(bytes.end(), b.begin(), b.end()); }
};

// ------------------------------------------------------------------------------------------------
// hybrid integer (inverse of j40__hybrid_int, j40.h:2313)

struct HybridCfg { int split_exp = 4, msb = 1, lsb = 0; };

struct HToken { uint32_t token; uint32_t extra; int nextra; };

inline HToken hybrid_encode(uint32_t v, const HybridCfg &c) {
	HToken t; uint32_t split = 1u << c.split_exp;
	if (v < split) { t.token = v; t.extra = 0; t.nextra = 0; return t; }
	int n = floor_lg(v), in_token = c.msb + c.lsb, midbits = n - in_token;
	t.token = split + (uint32_t) (((midbits - (c.split_exp - in_token)) << in_token) + (int) (((v >> (n - c.msb)) & ((1u << c.msb) - 1)) << c.lsb) + (int) (v & ((1u << c.lsb) - 1)));
	t.extra = (v >> c.lsb) & ((1u << midbits) - 1);
	t.nextra = midbits;
	return t;
}

inline void write_hybrid_cfg(BitWriter &bw, const HybridCfg &c, int log_alpha_size) {  // j40.h:2297
	bw.at_most(c.split_exp, log_alpha_size);
	if (c.split_exp != log_alpha_size) { bw.at_most(c.msb, c.split_exp); bw.at_most(c.lsb, c.split_exp - c.msb); }
}

// ------------------------------------------------------------------------------------------------
// rANS alias table exactly as the decoder constructs it (j40__init_alias_map, j40.h:2362), and
// the inverse slot map an encoder needs

struct AliasTable {
	int log_alpha = 6;
	std::vector<int> D;                       // 1 << log_alpha entries, sum 4096
	std::vector<std::vector<uint16_t>> inv;   // inv[s][r] = slot in [0,4096) that decodes to (s, r)
	void build() {
		const int table_size = 1 << log_alpha, log_bucket = 12 - log_alpha, bucket_size = 1 << log_bucket;
		std::vector<int> cutoff(table_size), off_or_next(table_size), symbol(table_size);
		int i, j, u = -1, o = -1;
		for (i = 0; i < table_size && !D[i]; ++i);
		for (j = i + 1; j < table_size && !D[j]; ++j);
		if (i < table_size && j >= table_size) {
			for (j = 0; j < table_size; ++j) { symbol[j] = i; off_or_next[j] = j << log_bucket; cutoff[j] = 0; }
		} else {
			for (i = 0; i < table_size; ++i) {
				cutoff[i] = D[i];
				if (cutoff[i] > bucket_size) { off_or_next[i] = o; o = i; }
				else if (cutoff[i] < bucket_size) { off_or_next[i] = u; u = i; }
				else { symbol[i] = i; off_or_next[i] = 0; }
			}
			while (o >= 0) {
				if (u < 0) die("alias: inconsistent histogram");
				int by = bucket_size - cutoff[u], tmp = off_or_next[u];
				cutoff[o] -= by; symbol[u] = o; off_or_next[u] = cutoff[o] - cutoff[u]; u = tmp;
				if (cutoff[o] < bucket_size) { tmp = off_or_next[o]; off_or_next[o] = u; u = o; o = tmp; }
				else if (cutoff[o] == bucket_size) { tmp = off_or_next[o]; off_or_next[o] = 0; symbol[o] = o; o = tmp; }
			}
		}
		inv.assign(table_size, {});
		for (int s = 0; s < table_size; ++s) inv[s].assign((size_t) D[s], 0xffff);
		for (int idx = 0; idx < 4096; ++idx) {  // decoder's bucket logic, j40.h:2450-2455
			int b = idx >> log_bucket, pos = idx & (bucket_size - 1);
			int s = pos < cutoff[b] ? b : symbol[b], offset = pos < cutoff[b] ? 0 : off_or_next[b];
			int r = offset + pos;
			if (s < 0 || s >= table_size || r < 0 || r >= D[s] || inv[s][r] != 0xffff) die("alias: slot map is not a bijection");
			inv[s][r] = (uint16_t) idx;
		}
	}
};

// ------------------------------------------------------------------------------------------------
// canonical prefix code (RFC 7932 section 3, read by j40__prefix_code_tree, j40.h:2049)

struct PrefixCode {
	int alphabet = 1;
	std::vector<int> len;        // per symbol, 0 = unused
	std::vector<uint32_t> code;  // canonical code, MSB-first value of `len` bits
	void assign_codes() {
		code.assign((size_t) alphabet, 0);
		int count[16] = {0}, next[16] = {0};
		for (int s = 0; s < alphabet; ++s) ++count[len[s]];
		count[0] = 0;
		uint32_t c = 0;
		for (int l = 1; l <= 15; ++l) { c = (c + (uint32_t) count[l - 1]) << 1; next[l] = (int) c; }
		for (int s = 0; s < alphabet; ++s) if (len[s]) code[s] = (uint32_t) next[len[s]]++;
	}
	void put_symbol(BitWriter &bw, int s) const {  // codes are read MSB-first one bit at a time
		for (int b = len[s] - 1; b >= 0; --b) bw.put((code[s] >> b) & 1, 1);
	}
};

// length-limited Huffman lengths (simple heuristic: build Huffman, then flatten while too deep)
inline std::vector<int> huffman_lengths(const std::vector<uint64_t> &freq, int maxlen) {
	int n = (int) freq.size();
	std::vector<int> len((size_t) n, 0);
	std::vector<int> used; for (int i = 0; i < n; ++i) if (freq[i]) used.push_back(i);
	if (used.empty()) return len;
	if (used.size() == 1) { len[used[0]] = 1; return len; }
	std::vector<uint64_t> f(freq);
	for (;;) {
		struct Node { uint64_t w; int l, r; };
		std::vector<Node> nodes; std::vector<int> heap;
		for (int s : used) { nodes.push_back({std::max<uint64_t>(f[s], 1), -1, s}); heap.push_back((int) nodes.size() - 1); }
		auto cmp = [&](int a, int b) { return nodes[a].w > nodes[b].w; };
		std::make_heap(heap.begin(), heap.end(), cmp);
		while (heap.size() > 1) {
			std::pop_heap(heap.begin(), heap.end(), cmp); int a = heap.back(); heap.pop_back();
			std::pop_heap(heap.begin(), heap.end(), cmp); int b = heap.back(); heap.pop_back();
			nodes.push_back({nodes[a].w + nodes[b].w, a, b}); heap.push_back((int) nodes.size() - 1);
			std::push_heap(heap.begin(), heap.end(), cmp);
		}
		int deepest = 0;
		std::vector<std::pair<int,int>> stack{{heap[0], 0}};
		while (!stack.empty()) {
			auto [id, d] = stack.back(); stack.pop_back();
			if (nodes[id].l < 0) { len[nodes[id].r] = d; deepest = std::max(deepest, d); }
			else { stack.push_back({nodes[id].l, d + 1}); stack.push_back({nodes[id].r, d + 1}); }
		}
		if (deepest <= maxlen) return len;
		for (int s : used) f[s] = f[s] / 2 + 1;  // flatten and retry
	}
}

// writes one prefix code tree for an alphabet of `count` symbols (count > 1) given per-symbol
// lengths; chooses the simple form for <= 4 used symbols (avoiding the NSYM=4/tree-select-0
// template the reference mis-orders) and the complex form otherwise
inline int &simple4_mode() { static int mode = 0; return mode; }   // generator option simple4=0|1|2, see write_prefix_tree
inline void write_prefix_tree(BitWriter &bw, PrefixCode &pc) {
	const int count = pc.alphabet;
	std::vector<int> used; for (int s = 0; s < count; ++s) if (pc.len[s]) used.push_back(s);
	const int symbits = ceil_lg((uint32_t) count);
	if (used.size() <= 4) {
		// simple code, j40.h:2084-2115; lengths are implied by the template
		int nsym = (int) used.size();
		bw.put(1, 2);
		bw.put((uint64_t) (nsym - 1), 2);
		if (nsym == 1) { bw.put((uint64_t) used[0], symbits); pc.len.assign((size_t) count, 0); pc.len[used[0]] = 0; pc.code.assign((size_t) count, 0); return; }
		if (nsym == 2) {
			bw.put((uint64_t) used[0], symbits); bw.put((uint64_t) used[1], symbits);
			pc.len.assign((size_t) count, 0); pc.len[used[0]] = pc.len[used[1]] = 1; pc.assign_codes(); return;
		}
		// order by given length so the most frequent symbol gets the short code
		std::stable_sort(used.begin(), used.end(), [&](int a, int b) { return pc.len[a] < pc.len[b]; });
		if (nsym == 3) {  // lengths 1,2,2; the two length-2 symbols are sorted by the reader
			for (int s : used) bw.put((uint64_t) s, symbits);
			pc.len.assign((size_t) count, 0); pc.len[used[0]] = 1; pc.len[used[1]] = pc.len[used[2]] = 2; pc.assign_codes(); return;
		}
		if (simple4_mode()) {
			// nsym == 4 with tree-select 0: four 2-bit codes over the sorted symbols. The reference decodes this template with symbol
			// k of the sorted four at the index made of the two bits in READ order (j40.h:2090, 2112), i.e. MSB-first code
			// (k & 1) << 1 | k >> 1 -- sorted symbols 1 and 2 swapped against RFC 7932's canonical code. Mode 1 writes what the
			// reference reads, mode 2 what an RFC 7932 encoder writes (tests/test_host.py documents the difference).
			std::sort(used.begin(), used.end());
			for (int s : used) bw.put((uint64_t) s, symbits);
			bw.put(0, 1);
			pc.len.assign((size_t) count, 0); pc.code.assign((size_t) count, 0);
			for (int k = 0; k < 4; ++k) { pc.len[used[(size_t) k]] = 2; pc.code[used[(size_t) k]] = simple4_mode() == 1 ? (uint32_t) (((k & 1) << 1) | (k >> 1)) : (uint32_t) k; }
			return;
		}
		// nsym == 4 with tree-select 1: lengths 1,2,3,3 (the two length-3 symbols are sorted)
		for (int s : used) bw.put((uint64_t) s, symbits);
		bw.put(1, 1);
		pc.len.assign((size_t) count, 0); pc.len[used[0]] = 1; pc.len[used[1]] = 2; pc.len[used[2]] = pc.len[used[3]] = 3; pc.assign_codes(); return;
	}
	// complex code, j40.h:2117-2177. layer 1 = code over code lengths 0..15 (no repeat codes used)
	std::vector<uint64_t> l1freq(18, 0);
	// symbols are listed until the Kraft sum is complete; trailing zeros are omitted
	int last = used.back();
	for (int s = 0; s <= last; ++s) ++l1freq[(size_t) pc.len[s]];
	std::vector<int> l1len = huffman_lengths(l1freq, 5);
	{ int nused = 0, only = -1; for (int i = 0; i < 18; ++i) if (l1len[i]) { ++nused; only = i; }
	  if (nused == 1) l1len[only] = 0; }  // a lone layer-1 symbol cannot fill the code space: handled below
	static const int L1ZIGZAG[18] = {1,2,3,4,0,5,17,6,16,7,8,9,10,11,12,13,14,15};
	// fixed layer-0 code for the values 0..5, bits in read order (derived from L0TABLE, j40.h:2063,
	// whose index is the next four bits read LSB-first)
	static const char *L0[6] = {"00", "1110", "110", "01", "10", "1111"};
	int nonzero_l1 = 0; for (int i = 0; i < 18; ++i) if (l1len[i]) ++nonzero_l1;
	PrefixCode l1; l1.alphabet = 18; l1.len = l1len;
	if (nonzero_l1 == 0) {
		// every listed symbol has the same length: give the lone layer-1 symbol a dummy sibling so
		// that the layer-1 lengths (1,1) fill the code space the reader checks (j40.h:2124)
		int only = -1; for (int i = 0; i < 18; ++i) if (l1freq[i]) only = i;
		int other = only == 0 ? 1 : 0;
		l1.len.assign(18, 0); l1.len[only] = 1; l1.len[other] = 1;
	}
	l1.assign_codes();
	bw.put(0, 2);  // hskip = 0
	{
		int total = 0;
		for (int i = 0; i < 18 && total < 32; ++i) {
			int l = l1.len[L1ZIGZAG[i]];
			for (const char *p = L0[l]; *p; ++p) bw.put((uint64_t) (*p - '0'), 1);
			if (l) total += 32 >> l;
		}
		if (total != 32) die("prefix: layer-1 lengths do not fill the code space");
	}
	{
		int total = 0;
		for (int s = 0; s < count && total < 32768; ++s) {
			int l = pc.len[s];
			l1.put_symbol(bw, l);
			if (l) total += 32768 >> l;
		}
		if (total != 32768) die("prefix: layer-2 lengths do not fill the code space");
	}
	pc.assign_codes();
}

// ------------------------------------------------------------------------------------------------
// entropy-coded streams: a code spec (inverse of j40__read_code_spec, j40.h:2711) whose histograms
// are gathered from the token streams that will use it, then per-stream encoding (inverse of
// j40__code, j40.h:2804)

struct Tok { uint32_t ctx; uint32_t value; };           // value = integer before hybrid-uint split
struct LzTok { uint32_t ctx; uint32_t value; uint32_t is_lz; uint32_t lz_len; uint32_t lz_dist_code; };

struct CodeSpecW {
	int num_dist = 1;                 // number of contexts (without the implicit LZ77 one)
	bool use_prefix = false;
	int log_alpha = 6;                // ANS only
	bool lz77 = false;
	int lz_min_symbol = 224, lz_min_length = 3;
	HybridCfg lz_len_cfg{0, 0, 0};    // read with log_alpha_size 8 (j40.h:2728)
	std::vector<uint8_t> cluster_map; // num_dist (+1 if lz77) entries
	int num_clusters = 1;
	std::vector<HybridCfg> cfg;       // per cluster
	std::vector<std::vector<uint64_t>> freq;  // per cluster token counts (gathered)
	std::vector<AliasTable> ans;      // per cluster (ANS)
	std::vector<PrefixCode> pfx;      // per cluster (prefix)
	int force_flat = 0;               // ANS: use the flat form for every cluster

	int alphabet_limit() const { return use_prefix ? (1 << 15) : (1 << log_alpha); }
	int total_dist() const { return num_dist + (lz77 ? 1 : 0); }

	void init(int ndist, const std::vector<uint8_t> &map, int nclusters) {
		num_dist = ndist; cluster_map = map; num_clusters = nclusters;
		if ((int) cluster_map.size() != total_dist()) die("cluster map size mismatch");
		cfg.assign((size_t) nclusters, HybridCfg{});
		freq.assign((size_t) nclusters, {});
	}
	void count_token(int cluster, uint32_t token) {
		auto &f = freq[(size_t) cluster];
		if (token >= f.size()) f.resize(token + 1, 0);
		++f[token];
	}
};

// normalise counts to sum 4096 with every seen symbol >= 1
inline std::vector<int> normalise_histogram(const std::vector<uint64_t> &f, int table_size) {
	std::vector<int> D((size_t) table_size, 0);
	uint64_t total = 0; int nz = 0;
	if ((int) f.size() > table_size) die("histogram wider than alphabet");
	for (size_t i = 0; i < f.size(); ++i) { total += f[i]; nz += f[i] != 0; }
	if (nz == 0) { D[0] = 4096; return D; }
	int64_t assigned = 0;
	for (size_t i = 0; i < f.size(); ++i) if (f[i]) {
		int d = (int) ((f[i] * 4096 + total / 2) / total);
		D[i] = d < 1 ? 1 : d; assigned += D[i];
	}
	while (assigned != 4096) {  // push the rounding error onto the currently largest entry
		size_t big = 0;
		for (size_t i = 1; i < f.size(); ++i) if (D[i] > D[big]) big = i;
		if (assigned < 4096) { D[big] += (int) (4096 - assigned); assigned = 4096; }
		else { int take = (int) std::min<int64_t>(assigned - 4096, D[big] - 1); if (take <= 0) die("cannot normalise histogram"); D[big] -= take; assigned -= take; }
	}
	return D;
}

// inverse of j40__ans_table (j40.h:2601)
inline void write_ans_histogram(BitWriter &bw, const std::vector<int> &D, int log_alpha, bool force_flat, int flat_alpha) {
	const int table_size = 1 << log_alpha;
	if (force_flat) { bw.put(2, 2); bw.u8(flat_alpha - 1); return; }
	std::vector<int> nzs; for (int i = 0; i < table_size; ++i) if (D[(size_t) i]) nzs.push_back(i);
	if (nzs.size() == 1) { bw.put(1, 2); bw.u8(nzs[0]); return; }                       // one entry
	if (nzs.size() == 2) { bw.put(3, 2); bw.u8(nzs[0]); bw.u8(nzs[1]); bw.put((uint64_t) D[(size_t) nzs[0]], 12); return; }  // two entries
	// general form: bit counts, no RLE; shift = 13 keeps every count exact
	bw.put(0, 2);
	bw.put(1, 1); bw.put(1, 1); bw.put(1, 1);  // len = 3
	bw.put(6, 3);                               // shift = 6 + 8 - 1 = 13
	int alpha_size = std::max(nzs.back() + 1, 3);
	bw.u8(alpha_size - 3);
	static const char *LOGCOUNT[14] = {"10001", "1101", "1111", "1100", "1001", "1110", "001", "010", "101", "011", "000", "100001", "1000000", "1000001"};
	std::vector<int> k((size_t) alpha_size);
	int kmax = -1, omit = -1;
	for (int i = 0; i < alpha_size; ++i) { k[(size_t) i] = D[(size_t) i] ? floor_lg((uint32_t) D[(size_t) i]) + 1 : 0; if (k[(size_t) i] > kmax) { kmax = k[(size_t) i]; omit = i; } }
	for (int i = 0; i < alpha_size; ++i) for (const char *p = LOGCOUNT[k[(size_t) i]]; *p; ++p) bw.put((uint64_t) (*p - '0'), 1);
	for (int i = 0; i < alpha_size; ++i) {
		if (i == omit || k[(size_t) i] < 2) continue;
		bw.put((uint64_t) (D[(size_t) i] - (1 << (k[(size_t) i] - 1))), k[(size_t) i] - 1);
	}
}

struct StreamEncoder;

// cluster map (inverse of j40__cluster_map, j40.h:2526)
void write_cluster_map(BitWriter &bw, const std::vector<uint8_t> &map, int num_clusters);

// finalises tables from the gathered counts and writes the spec
inline void write_code_spec(BitWriter &bw, CodeSpecW &spec) {
	bw.put(spec.lz77 ? 1 : 0, 1);
	if (spec.lz77) {
		bw.u32(spec.lz_min_symbol, 224, 0, 512, 0, 4096, 0, 8, 15);
		bw.u32(spec.lz_min_length, 3, 0, 4, 0, 5, 2, 9, 8);
		write_hybrid_cfg(bw, spec.lz_len_cfg, 8);
	}
	if (spec.total_dist() > 1) write_cluster_map(bw, spec.cluster_map, spec.num_clusters);
	bw.put(spec.use_prefix ? 1 : 0, 1);
	if (spec.use_prefix) {
		for (int c = 0; c < spec.num_clusters; ++c) write_hybrid_cfg(bw, spec.cfg[(size_t) c], 15);
		spec.pfx.assign((size_t) spec.num_clusters, PrefixCode{});
		for (int c = 0; c < spec.num_clusters; ++c) {
			auto &f = spec.freq[(size_t) c];
			int count = 1; for (size_t i = 0; i < f.size(); ++i) if (f[i]) count = (int) i + 1;
			spec.pfx[(size_t) c].alphabet = count;
			if (count == 1) { bw.put(0, 1); }
			else { int n = floor_lg((uint32_t) (count - 1)); bw.put(1, 1); bw.put((uint64_t) n, 4); bw.put((uint64_t) (count - 1 - (1 << n)), n); }
		}
		for (int c = 0; c < spec.num_clusters; ++c) {
			PrefixCode &pc = spec.pfx[(size_t) c];
			if (pc.alphabet == 1) { pc.len.assign(1, 0); pc.code.assign(1, 0); continue; }
			std::vector<uint64_t> f(spec.freq[(size_t) c]); f.resize((size_t) pc.alphabet, 0);
			pc.len = huffman_lengths(f, 15);
			write_prefix_tree(bw, pc);
		}
	} else {
		bw.put((uint64_t) (spec.log_alpha - 5), 2);
		for (int c = 0; c < spec.num_clusters; ++c) write_hybrid_cfg(bw, spec.cfg[(size_t) c], spec.log_alpha);
		spec.ans.assign((size_t) spec.num_clusters, AliasTable{});
		for (int c = 0; c < spec.num_clusters; ++c) {
			AliasTable &t = spec.ans[(size_t) c];
			t.log_alpha = spec.log_alpha;
			const int table_size = 1 << spec.log_alpha;
			int flat_alpha = 0;
			if (spec.force_flat) {
				auto &f = spec.freq[(size_t) c];
				flat_alpha = 1; for (size_t i = 0; i < f.size(); ++i) if (f[i]) flat_alpha = (int) i + 1;
				t.D.assign((size_t) table_size, 0);
				int d = 4096 / flat_alpha, bias = 4096 % flat_alpha;
				for (int i = 0; i < flat_alpha; ++i) t.D[(size_t) i] = d + (i < bias);
			} else {
				t.D = normalise_histogram(spec.freq[(size_t) c], table_size);
			}
			write_ans_histogram(bw, t.D, spec.log_alpha, spec.force_flat != 0, flat_alpha);
			t.build();
		}
	}
}

// one entropy-coded stream. Tokens are appended in decode order, then flush() writes them.
struct StreamEncoder {
	const CodeSpecW *spec;
	struct Item { uint32_t cluster, token, extra; uint8_t nextra; };
	std::vector<Item> items;
	explicit StreamEncoder(const CodeSpecW &s) : spec(&s) {}
	void add(uint32_t ctx, uint32_t value) {
		uint32_t cl = spec->cluster_map[ctx];
		HToken t = hybrid_encode(value, spec->cfg[cl]);
		items.push_back({cl, t.token, t.extra, (uint8_t) t.nextra});
	}
	// LZ77 copy: the length token is coded with the cluster of `ctx` (j40.h:2822-2825), the
	// distance token with the cluster of the last context (j40.h:2824-2827)
	void add_lz(uint32_t ctx, uint32_t copy_len, uint32_t dist_code) {
		uint32_t cl = spec->cluster_map[ctx];
		HToken t = hybrid_encode(copy_len - (uint32_t) spec->lz_min_length, spec->lz_len_cfg);
		items.push_back({cl, t.token + (uint32_t) spec->lz_min_symbol, t.extra, (uint8_t) t.nextra});
		uint32_t lzcl = spec->cluster_map[(size_t) spec->total_dist() - 1];
		HToken d = hybrid_encode(dist_code, spec->cfg[lzcl]);
		items.push_back({lzcl, d.token, d.extra, (uint8_t) d.nextra});
	}
	void flush(BitWriter &bw) {
		if (spec->use_prefix) {
			for (const Item &it : items) {
				const PrefixCode &pc = spec->pfx[it.cluster];
				if (pc.alphabet > 1) pc.put_symbol(bw, (int) it.token);
				bw.put(it.extra, it.nextra);
			}
		} else {
			// rANS, encoded in reverse (inverse of j40__ans_code, j40.h:2441)
			std::vector<uint32_t> word(items.size(), 0xffffffffu);
			uint32_t x = 0x130000;
			for (size_t k = items.size(); k-- > 0; ) {
				const AliasTable &t = spec->ans[items[k].cluster];
				uint32_t d = (uint32_t) t.D[items[k].token];
				if (!d) die("rANS: symbol with zero probability");
				if ((x >> 20) >= d) { word[k] = x & 0xffff; x >>= 16; }
				x = ((x / d) << 12) + t.inv[items[k].token][x % d];
			}
			bw.put(x & 0xffff, 16); bw.put(x >> 16, 16);
			for (size_t k = 0; k < items.size(); ++k) {
				if (word[k] != 0xffffffffu) bw.put(word[k], 16);
				bw.put(items[k].extra, items[k].nextra);
			}
		}
		items.clear();
	}
};

// gather pass helper: count all items of a stream into the spec's frequency tables
inline void count_stream(CodeSpecW &spec, const StreamEncoder &enc) {
	for (const auto &it : enc.items) spec.count_token((int) it.cluster, it.token);
}

inline void write_cluster_map(BitWriter &bw, const std::vector<uint8_t> &map, int num_clusters) {
	int nbits = ceil_lg((uint32_t) num_clusters);
	if (nbits <= 3) {
		bw.put(1, 1); bw.put((uint64_t) nbits, 2);
		for (uint8_t m : map) bw.put(m, nbits);
		return;
	}
	// general form: a nested single-context ANS stream with a flat distribution, no MTF
	bw.put(0, 1);  // !is_simple
	bw.put(0, 1);  // use_mtf = 0
	CodeSpecW nested;
	nested.init(1, std::vector<uint8_t>{0}, 1);
	nested.use_prefix = false;
	nested.log_alpha = num_clusters <= 64 ? 6 : 8;
	nested.cfg[0] = HybridCfg{nested.log_alpha, 0, 0};
	nested.force_flat = 1;
	nested.freq[0].assign((size_t) num_clusters, 1);
	write_code_spec(bw, nested);
	StreamEncoder enc(nested);
	for (uint8_t m : map) enc.add(0, m);
	enc.flush(bw);
}

inline bool write_file(const char *path, const std::vector<uint8_t> &bytes) {
	FILE *fp = fopen(path, "wb");
	if (!fp) return false;
	size_t n = fwrite(bytes.data(), 1, bytes.size(), fp);
	fclose(fp);
	return n == bytes.size();
}

// ICC stream as the reference reads it (j40.h:3351-3393): enc_size, a 41-context code spec, the output size as a varint
// coded with context 0, then the remaining enc_size - (varint bytes) command/data bytes with the previous-two-bytes
// context model. The reference decodes and discards the bytes, so their content is free.
template <typename RNG> static inline void write_icc_stream(BitWriter &bw, RNG &rng, int enc_size) {
	CodeSpecW spec;
	std::vector<uint8_t> map(41);
	for (int i = 0; i < 41; ++i) map[(size_t) i] = (uint8_t) (i == 0 ? 0 : 1 + i % 2);
	spec.init(41, map, 3);
	spec.log_alpha = 8;
	for (auto &c : spec.cfg) c = HybridCfg{4, 2, 0};
	StreamEncoder enc(spec);
	uint64_t index = 0;
	{   // output_size = enc_size (every byte produces at least one output byte in the reference's model)
		uint64_t v = (uint64_t) enc_size;
		do { uint32_t b = (uint32_t) (v & 0x7f); v >>= 7; if (v) b |= 0x80; enc.add(0, b); ++index; } while (v);
	}
	int byte = 0, prev = 0, pprev = 0;
	for (; index < (uint64_t) enc_size; ++index) {
		pprev = prev; prev = byte;
		int ctx = 0;
		if (index > 128) {
			if (prev < 16) ctx = prev < 2 ? prev + 3 : 5;
			else if (prev > 240) ctx = 6 + (prev == 255);
			else if (97 <= (prev | 32) && (prev | 32) <= 122) ctx = 1;
			else if (prev == 44 || prev == 46 || (48 <= prev && prev < 58)) ctx = 2;
			else ctx = 8;
			if (pprev < 16) ctx += 2 * 8;
			else if (pprev > 240) ctx += 3 * 8;
			else if (97 <= (pprev | 32) && (pprev | 32) <= 122) ctx += 0 * 8;
			else if (pprev == 44 || pprev == 46 || (48 <= pprev && pprev < 58)) ctx += 1 * 8;
			else ctx += 4 * 8;
		}
		const uint32_t r = rng.below(8);
		byte = r < 3 ? (int) rng.below(256) : r < 5 ? 'a' + (int) rng.below(26) : r == 5 ? '0' + (int) rng.below(10) : r == 6 ? 0 : 255;
		enc.add((uint32_t) ctx, (uint32_t) byte);
	}
	count_stream(spec, enc);
	bw.u64((uint64_t) enc_size);
	write_code_spec(bw, spec);
	enc.flush(bw);
}

} // namespace synth
