// inflate_lanes.hpp - raw DEFLATE (RFC 1951), ONE BGZF BLOCK PER LANE: 64 independent streams per wavefront, every lane a plain serial decoder with its
// own code tables in LDS.  The second inflater of bgzf.hip (round 6): BGZF cuts a file into independent 64 KiB blocks, a BAM file has 10^5 of them, so the
// parallelism is between the blocks and nothing inside a block has to be speculated (inflate_core.hpp decodes one block per wavefront and speculates on 64
// code starts per step: 470 instructions per 7 tokens).  A lane does per loop trip what a CPU decoder does per token; what makes it fast on a SIMT machine is
// that every trip is the SAME instruction stream for all lanes:
//   * one trip = at most one token decoded (literal, or length + distance) AND one step of the lane's copy engine (the 4 bytes of a match loaded in the trip
//     before are stored, the next 4 are requested) - a short match costs one trip like a literal, its load has a whole trip to come back;
//   * block headers (code lengths, table build) are serial per lane and long: a lane that reaches one parks until INFL_HDR_BATCH lanes of its wave wait
//     (or nobody decodes any more), then they read their headers together;
//   * code tables: a root table of 2^INFL_LR entries and second-level tables for the longer codes in a fixed budget; a block whose codes do not fit, a
//     stored block or anything irregular makes the lane give up - the block is then redone by the wave-per-block decoder, which also owns the error codes.
// The same source compiles for the host (INFL_HOST: tools/inflate_lanes_host_test.cpp, every lane run on its own, against zlib).
// Replaces: zlib's inflate() under htslib's bgzf_read_block (pysam.AlignmentFile, /root/reference/src/svim/SVIM_COLLECT.py:132-137).
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef INFL_LR
#define INFL_LR 6              /* root bits of the literal/length table */
#endif
#ifndef INFL_LSUB
#define INFL_LSUB 132          /* entries for its second- and third-level tables */
#endif
#ifndef INFL_DR
#define INFL_DR 5
#endif
#ifndef INFL_DSUB
#define INFL_DSUB 56
#endif
#ifndef INFL_HDR_BATCH
#define INFL_HDR_BATCH 8
#endif

// per-lane scratch, byte offsets (u16 tables first)
#define INFL_LT_N ((1 << INFL_LR) + INFL_LSUB)
#define INFL_DT_N ((1 << INFL_DR) + INFL_DSUB)
#define INFL_OFF_LT 0
#define INFL_OFF_DT (2 * INFL_LT_N)
#define INFL_OFF_CNT (INFL_OFF_DT + 2 * INFL_DT_N)       /* u16 count[16], u16 next[16] */
#define INFL_BYTES (INFL_OFF_CNT + 64)
// while a header is read the literal/length region (built last) holds the 320 code lengths, 4 bits each, and the 7-bit table of the code-length code; before the
// literal/length table is built over them the lengths move to the lane's 160 bytes of GLOBAL scratch (read back 8 lengths per load, three sequential passes)
#define INFL_OFF_LENS INFL_OFF_LT
#define INFL_OFF_CL (INFL_OFF_LT + 160)
#define INFL_LENS_WORDS 40                                 /* dwords of global scratch per lane */
static_assert(2 * INFL_LT_N >= 288, "the header needs 288 bytes of the literal/length region");
static_assert(15 - INFL_LR - 4 <= 7 && 15 - INFL_DR - 4 <= 7, "third-level index bits are a 3-bit field");
static_assert((INFL_BYTES & 3) == 0, "scratch is a whole number of dwords");
#define INFL_STRIDE (INFL_BYTES + (((INFL_BYTES >> 2) & 1) ? 0 : 4))   /* distance of two lanes' scratch in LDS: an odd number of dwords, so that the lanes of a wave
                                                                          that touch the same entry of their tables (the table builds) use 64 different banks */

#define INFL_ST_DECODE 0u
#define INFL_ST_HEADER 1u
#define INFL_ST_DONE 2u
#define INFL_ST_FAIL 3u

#ifdef INFL_HOST
#define INFL_FN static inline
#define INFL_MFN inline
static inline uint32_t infl_bitrev(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (x & 1u); x >>= 1; } return r; }
static inline uint32_t infl_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u)); }
#else
#define INFL_FN __device__ __forceinline__
#define INFL_MFN __device__ __forceinline__
static __device__ __forceinline__ uint32_t infl_bitrev(uint32_t x) { return __builtin_bitreverse32(x); }
static __device__ __forceinline__ uint32_t infl_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
#endif
#ifndef INFL_STAT
#define INFL_STAT(what, n)
#endif
#ifndef INFL_FAIL
#define INFL_FAIL(L, site) (L).state = INFL_ST_FAIL          /* the host test counts the sites */
#endif

INFL_FN uint64_t infl_ld64(const uint8_t* p) { uint64_t v; memcpy(&v, __builtin_assume_aligned(p, 8), 8); return v; }
typedef uint32_t __attribute__((aligned(1), may_alias)) infl_u32u;
INFL_FN uint32_t infl_ld32u(const uint8_t* p) { return *reinterpret_cast<const infl_u32u*>(p); }

#ifndef INFL_DEPTH
#define INFL_DEPTH 3           /* trips between the load of a match chunk and its store */
#endif
struct InflLane {
    // input: 64-bit words of the stream from the 8-byte boundary in front of the payload; w0:w1 hold the bits being decoded, w2, w3 and pw are the words after
    // them.  A trip moves the window by one word at most (48 bits per token); pw is loaded at the END of EVERY trip (the word behind w3, whether the window moved
    // or not: an unconditional load is one the compiler can count - see infl_step), and nothing reads it before the window moves in a later trip
    uint64_t w0, w1, w2, w3, pw;
    uint32_t bo;                       // bit offset of the next code in w0:w1; < 64 when a trip begins
    const uint8_t* in0;                // aligned start
    uint32_t in_at, in_lim, in_bits;   // byte offset of pw; loads stop at in_lim; last valid bit of the payload (from in0)
    // output
    uint8_t* out; uint32_t pos, cap;
    // copy engine (infl_step): the match being loaded (c_rem bytes still to request for out[c_dst ...]), the match decoded behind it (m_len != 0: waiting for the
    // engine), and INFL_DEPTH chunk slots: s_data[k] was requested in a trip of parity k and is stored INFL_DEPTH trips later, s_n[k] = bytes | (distance < 4 ? distance << 4 : 0)
    uint32_t c_rem, c_dst, c_dist;
    uint32_t m_len, m_dst, m_dist;
    uint32_t s_data[INFL_DEPTH], s_dst[INFL_DEPTH], s_n[INFL_DEPTH], n_pend, p_min;     // p_min: no pending chunk lies below this output position
    uint32_t state, fin;
    uint8_t* S;                        // this lane's scratch (LDS)
    uint32_t* G; uint32_t g_stride;    // this lane's INFL_LENS_WORDS dwords of global scratch: word k at G[k * g_stride]; the last word is where idle stores / loads go
};

INFL_FN uint32_t infl_peek(const InflLane& L, uint32_t b) {              // 32 bits from bit b (< 64) of w0:w1
    const uint32_t a = (uint32_t)L.w0, m = (uint32_t)(L.w0 >> 32), c = (uint32_t)L.w1;
    const bool up = (b & 32u) != 0u;
    return infl_alignbit(up ? c : m, up ? m : a, b);
}
INFL_FN uint64_t infl_word_at(const InflLane& L, uint32_t at) {          // a load in any case (what lies behind the payload belongs to the buffer: see infl_init)
    return infl_ld64(L.in0 + (at < L.in_lim ? at : L.in_lim));
}
INFL_FN void infl_shift(InflLane& L) {
    if (L.bo >= 64u) { L.w0 = L.w1; L.w1 = L.w2; L.w2 = L.w3; L.w3 = L.pw; L.bo -= 64u; L.in_at += 8u; }
}
// position the window on absolute bit `bit` of the stream (from in0)
INFL_FN void infl_seek(InflLane& L, uint32_t bit) {
    const uint32_t at = (bit >> 6) << 3;
    L.bo = bit & 63u;
    L.w0 = infl_word_at(L, at); L.w1 = infl_word_at(L, at + 8u); L.w2 = infl_word_at(L, at + 16u); L.w3 = infl_word_at(L, at + 24u); L.pw = infl_word_at(L, at + 32u);
    L.in_at = at + 32u;
}
INFL_FN uint32_t infl_bitpos(const InflLane& L) { return 8u * (L.in_at - 32u) + L.bo; }

INFL_FN void infl_init(InflLane& L, const uint8_t* payload, uint32_t in_bytes, uint8_t* out, uint32_t cap, uint8_t* S, uint32_t* G, uint32_t g_stride) {
    const uint32_t skip = (uint32_t)(reinterpret_cast<uintptr_t>(payload) & 7u);
    L.in0 = payload - skip;
    L.in_bits = 8u * (skip + in_bytes);
    L.in_lim = (skip + in_bytes + 7u) & ~7u;                   // loads behind the payload all read this word: 8 bytes behind the last word of a payload must exist
    L.out = out; L.pos = 0; L.cap = cap;
    L.c_rem = L.c_dst = L.c_dist = 0; L.m_len = L.m_dst = L.m_dist = 0; L.n_pend = 0; L.p_min = 0;
    for (int k = 0; k < INFL_DEPTH; k++) { L.s_data[k] = 0; L.s_dst[k] = 0; L.s_n[k] = 0; }
    L.state = INFL_ST_HEADER; L.fin = 0;
    L.S = S; L.G = G; L.g_stride = g_stride;
    infl_seek(L, 8u * skip);
}

// ---- block header ---------------------------------------------------------------------------------------------------------------------------------
INFL_FN uint32_t infl_len_get(const uint8_t* S, uint32_t i) { return (uint32_t)(S[INFL_OFF_LENS + (i >> 1)] >> ((i & 1u) * 4u)) & 15u; }
INFL_FN void infl_len_set(uint8_t* S, uint32_t i, uint32_t v) {
    uint8_t& b = S[INFL_OFF_LENS + (i >> 1)];
    b = (uint8_t)((i & 1u) ? ((b & 0x0fu) | (v << 4)) : ((b & 0xf0u) | v));
}
// table entries (u16): code length in the low 4 bits, payload above.  Literal/length table: payload < 256 literal, 256 end of block, 0x800 | extra bits << 8 |
// (base length - 3) a length symbol, 0x200 a symbol that must not occur; length field 0 with a payload: second level, payload = index bits << 8 | offset; 0: no code
INFL_FN uint32_t infl_entry_lit(uint32_t sym, uint32_t len) {
    uint32_t pay;
    if (sym <= 256u) pay = sym;
    else if (sym > 285u) pay = 0x200u;
    else {
        const uint32_t s = sym - 257u;
        const uint32_t xb = (s < 8u || s == 28u) ? 0u : (s - 4u) >> 2;
        const uint32_t base = s < 8u ? 3u + s : (s == 28u ? 258u : 3u + ((4u + (s & 3u)) << xb));
        pay = 0x800u | (xb << 8) | (base - 3u);
    }
    return (pay << 4) | len;
}
INFL_FN uint32_t infl_entry_dist(uint32_t sym, uint32_t len) { return ((sym < 30u ? sym : 0x200u) << 4) | len; }

// code lengths [first, first + n) -> a three-level table at tab (u16): the root is indexed by R bits; a root slot shared by longer codes points to a second-level
// table indexed by the next min(INFL_L2, longest - R) bits; a second-level slot shared by still longer codes points to a third-level table indexed by the rest
// (flat second-level tables would need 2^(15 - R) entries under the slot of the longest codes: the budget of 64 lanes' tables in LDS does not have them).
// Markers while building: low nibble 0, longest code length through the slot << 12.  Returns 0 = built, 1 = over-subscribed code, 2 = budget exceeded
#ifndef INFL_L2
#define INFL_L2 4
#endif
struct InflLensLds {                       // the lengths where the header left them
    const uint8_t* S; uint32_t first;
    INFL_MFN void rewind() {}
    INFL_MFN uint32_t next(uint32_t i) { return infl_len_get(S, first + i); }
};
struct InflLensGlobal {                    // the lane's global copy: symbols 0, 1, 2 ... in order, one load per 8 of them
    const uint32_t* G; uint32_t stride, w;
    INFL_MFN void rewind() { w = 0; }
    INFL_MFN uint32_t next(uint32_t i) { if ((i & 7u) == 0u) w = G[(i >> 3) * stride]; const uint32_t l = w & 15u; w >>= 4; return l; }
};
template <bool LIT, class Lens>
INFL_FN int infl_build(uint8_t* S, Lens lens, uint32_t n, uint16_t* tab, const uint32_t R, const uint32_t sub_cap) {
    uint16_t* cnt = reinterpret_cast<uint16_t*>(S + INFL_OFF_CNT);
    uint16_t* nxt = cnt + 16;
    for (uint32_t l = 0; l < 16u; l++) cnt[l] = 0;
    lens.rewind();
    for (uint32_t i = 0; i < n; i++) cnt[lens.next(i)]++;
    cnt[0] = 0;
    int left = 1;
    uint32_t code = 0, longs = 0, deep = 0;
    for (uint32_t l = 1; l < 16u; l++) {
        left = 2 * left - (int)cnt[l];
        if (left < 0) return 1;
        code = (code + cnt[l - 1u]) << 1;
        nxt[l] = (uint16_t)code;
        if (l > R) longs += cnt[l];
        if (l > R + INFL_L2) deep += cnt[l];
    }
    const uint32_t root_n = 1u << R;
    uint16_t* sub = tab + root_n;
    for (uint32_t k = 0; k < root_n; k++) tab[k] = 0;
    // pass 1: the short codes fill the root, a longer code leaves its length in its root slot
    lens.rewind();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = lens.next(i);
        if (!l) continue;
        const uint32_t c = nxt[l]++;
        const uint32_t rev = infl_bitrev(c) >> (32u - l);
        if (l <= R) {
            const uint16_t e = (uint16_t)(LIT ? infl_entry_lit(i, l) : infl_entry_dist(i, l));
            for (uint32_t k = rev; k < root_n; k += 1u << l) tab[k] = e;
        } else {
            const uint32_t p = rev & (root_n - 1u);
            if ((uint32_t)tab[p] < (l << 12)) tab[p] = (uint16_t)(l << 12);
        }
    }
    if (!longs) { INFL_STAT(LIT ? 0 : 1, 0); return 0; }
    uint32_t off = 0;
    for (uint32_t p = 0; p < root_n; p++) {
        const uint32_t e = tab[p];
        if (e == 0u || (e & 15u) != 0u) continue;
        const uint32_t rest = (e >> 12) - R, bits = rest < INFL_L2 ? rest : INFL_L2, size = 1u << bits;
        if (off + size > sub_cap) return 2;
        for (uint32_t k = 0; k < size; k++) sub[off + k] = 0;
        tab[p] = (uint16_t)(((bits << 8) | off) << 4);
        off += size;
    }
    // pass 2: the codes of up to R + INFL_L2 bits into the second level; a longer one leaves its length in its second-level slot
    code = 0;
    for (uint32_t l = 1; l < 16u; l++) { code = (code + cnt[l - 1u]) << 1; nxt[l] = (uint16_t)code; }
    lens.rewind();
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = lens.next(i);
        if (!l) continue;
        const uint32_t c = nxt[l]++;
        if (l <= R) continue;
        const uint32_t rev = infl_bitrev(c) >> (32u - l);
        const uint32_t e = tab[rev & (root_n - 1u)] >> 4, bits = e >> 8, o = e & 255u, rest = l - R;
        if (rest <= INFL_L2) {
            const uint16_t v = (uint16_t)(LIT ? infl_entry_lit(i, l) : infl_entry_dist(i, l));
            for (uint32_t k = rev >> R; k < (1u << bits); k += 1u << rest) sub[o + k] = v;
        } else {
            uint16_t& slot = sub[o + ((rev >> R) & ((1u << INFL_L2) - 1u))];
            if ((uint32_t)slot < (l << 12)) slot = (uint16_t)(l << 12);
        }
    }
    if (deep) {
        const uint32_t l2_end = off;
        for (uint32_t q = 0; q < l2_end; q++) {
            const uint32_t e = sub[q];
            if (e == 0u || (e & 15u) != 0u) continue;
            const uint32_t bits = (e >> 12) - R - INFL_L2, size = 1u << bits;
            if (off + size > sub_cap) return 2;
            for (uint32_t k = 0; k < size; k++) sub[off + k] = 0;
            sub[q] = (uint16_t)(((bits << 8) | off) << 4);
            off += size;
        }
        code = 0;
        for (uint32_t l = 1; l < 16u; l++) { code = (code + cnt[l - 1u]) << 1; nxt[l] = (uint16_t)code; }
        lens.rewind();
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t l = lens.next(i);
            if (!l) continue;
            const uint32_t c = nxt[l]++;
            if (l <= R + INFL_L2) continue;
            const uint32_t rev = infl_bitrev(c) >> (32u - l);
            const uint32_t e2 = tab[rev & (root_n - 1u)] >> 4;
            const uint32_t e3 = sub[(e2 & 255u) + ((rev >> R) & ((1u << INFL_L2) - 1u))] >> 4, bits = e3 >> 8, o = e3 & 255u, rest = l - R - INFL_L2;
            const uint16_t v = (uint16_t)(LIT ? infl_entry_lit(i, l) : infl_entry_dist(i, l));
            for (uint32_t k = rev >> (R + INFL_L2); k < (1u << bits); k += 1u << rest) sub[o + k] = v;
        }
    }
    INFL_STAT(LIT ? 0 : 1, off);
    return 0;
}
// the entry of the code at the low end of x: root, and up to two more levels
INFL_FN uint32_t infl_lookup(const uint16_t* tab, const uint32_t R, uint32_t x) {
    uint32_t e = tab[x & ((1u << R) - 1u)];
    if ((e & 15u) == 0u && e != 0u) {
        const uint32_t p = e >> 4;
        e = tab[(1u << R) + (p & 255u) + ((x >> R) & ((1u << (p >> 8)) - 1u))];
        if ((e & 15u) == 0u && e != 0u) {
            const uint32_t q = e >> 4;
            e = tab[(1u << R) + (q & 255u) + ((x >> (R + INFL_L2)) & ((1u << (q >> 8)) - 1u))];
        }
    }
    return e;
}

// serial bit reader of the header: two words of the stream in registers, reloaded when the position leaves the first
struct InflBits { const uint8_t* in0; uint32_t bit, lim, wi; uint64_t lo, hi; };
INFL_FN uint64_t infl_hword(const InflBits& b, uint32_t wi) { const uint32_t at = wi << 3; return infl_ld64(b.in0 + (at < b.lim ? at : b.lim)); }
INFL_FN uint32_t infl_hpeek(InflBits& b) {                              // 32 bits from b.bit
    const uint32_t wi = b.bit >> 6, s = b.bit & 63u;
    if (wi != b.wi) { b.lo = wi == b.wi + 1u ? b.hi : infl_hword(b, wi); b.hi = infl_hword(b, wi + 1u); b.wi = wi; }
    return (uint32_t)(s ? (b.lo >> s) | (b.hi << (64u - s)) : b.lo);
}

// the lane stands on a block header: read it, build the tables.  Leaves the lane in DECODE (window on the first code) or FAIL
INFL_FN void infl_header(InflLane& L) {
    uint8_t* S = L.S;
    InflBits hb{L.in0, infl_bitpos(L), L.in_lim, 0xfffffff0u, 0ull, 0ull};
    uint32_t x = infl_hpeek(hb);
    L.fin = x & 1u;
    const uint32_t type = (x >> 1) & 3u;
    hb.bit += 3u;
    uint32_t nlen, ndist;
    if (type == 1u) {
        nlen = 288u; ndist = 32u;
        for (uint32_t i = 0; i < 288u; i++) infl_len_set(S, i, i < 144u ? 8u : (i < 256u ? 9u : (i < 280u ? 7u : 8u)));
        for (uint32_t i = 0; i < 32u; i++) infl_len_set(S, 288u + i, 5u);
    } else if (type == 2u) {
        x >>= 3;
        nlen = (x & 31u) + 257u; ndist = ((x >> 5) & 31u) + 1u;
        const uint32_t ncode = ((x >> 10) & 15u) + 4u;
        hb.bit += 14u;
        if (nlen > 286u || ndist > 30u) { INFL_FAIL(L, 1); return; }
        // the code-length code: 19 symbols of 3 bits each in a fixed order -> a 7-bit table of (length << 5 | symbol)
        uint8_t* cl = S + INFL_OFF_CL;
        uint16_t* cnt = reinterpret_cast<uint16_t*>(S + INFL_OFF_CNT);
        uint16_t* nxt = cnt + 16;
        uint32_t cll[3] = {0u, 0u, 0u};                                   // 19 lengths of 3 bits, by symbol
        for (uint32_t k = 0; k < ncode; k++) {
            if ((k & 7u) == 0u) x = infl_hpeek(hb);
            const uint32_t v = (x >> (3u * (k & 7u))) & 7u;
            if ((k & 7u) == 7u || k + 1u == ncode) hb.bit += 3u * ((k & 7u) + 1u);
            // the fixed order of RFC 1951 3.2.7 (16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15), 5 bits each
            const uint32_t sym = (uint32_t)((k < 12u ? 0x22caa324e804a30ull >> (5u * k) : 0x3c2e1346cull >> (5u * (k - 12u))) & 31ull);
            cll[sym / 8u] |= v << (3u * (sym % 8u));
        }
        for (uint32_t l = 0; l < 8u; l++) cnt[l] = 0;
        for (uint32_t s = 0; s < 19u; s++) cnt[(cll[s / 8u] >> (3u * (s % 8u))) & 7u]++;
        cnt[0] = 0;
        int left = 1; uint32_t code = 0;
        for (uint32_t l = 1; l < 8u; l++) { left = 2 * left - (int)cnt[l]; code = (code + cnt[l - 1u]) << 1; nxt[l] = (uint16_t)code; }
        if (left != 0) { INFL_FAIL(L, 2); return; }                // zlib wants a complete code here
        for (uint32_t k = 0; k < 128u; k++) cl[k] = 0;
        for (uint32_t s = 0; s < 19u; s++) {
            const uint32_t l = (cll[s / 8u] >> (3u * (s % 8u))) & 7u;
            if (!l) continue;
            const uint32_t c = nxt[l]++;
            const uint32_t rev = infl_bitrev(c) >> (32u - l);
            for (uint32_t k = rev; k < 128u; k += 1u << l) cl[k] = (uint8_t)((l << 5) | s);
        }
        // the code lengths themselves
        const uint32_t total = nlen + ndist;
        uint32_t i = 0, prev = 0;
        while (i < total) {
            x = infl_hpeek(hb);
            const uint32_t e = cl[x & 127u], l = e >> 5, s = e & 31u;
            if (!l) { INFL_FAIL(L, 3); return; }
            x >>= l;
            if (s < 16u) { infl_len_set(S, i, s); prev = s; i++; hb.bit += l; continue; }
            uint32_t rep, val = 0;
            if (s == 16u) { if (i == 0u) { INFL_FAIL(L, 4); return; } val = prev; rep = 3u + (x & 3u); hb.bit += l + 2u; }
            else if (s == 17u) { rep = 3u + (x & 7u); hb.bit += l + 3u; }
            else { rep = 11u + (x & 127u); hb.bit += l + 7u; }
            if (i + rep > total) { INFL_FAIL(L, 5); return; }
            for (uint32_t r = 0; r < rep; r++) infl_len_set(S, i + r, val);
            i += rep; prev = val;
        }
        if (infl_len_get(S, 256u) == 0u) { INFL_FAIL(L, 6); return; }
    } else { INFL_FAIL(L, 7); return; }                           // stored blocks (and the reserved type): the other decoder
    if (infl_build<false>(S, InflLensLds{S, nlen}, ndist, reinterpret_cast<uint16_t*>(S + INFL_OFF_DT), INFL_DR, INFL_DSUB)) { INFL_FAIL(L, 8); return; }
    {
        const uint32_t* ls = reinterpret_cast<const uint32_t*>(S + INFL_OFF_LENS);
        for (uint32_t k = 0; k < (nlen + 7u) / 8u; k++) L.G[k * L.g_stride] = ls[k];
    }
    if (infl_build<true>(S, InflLensGlobal{L.G, L.g_stride, 0u}, nlen, reinterpret_cast<uint16_t*>(S + INFL_OFF_LT), INFL_LR, INFL_LSUB)) { INFL_FAIL(L, 8); return; }
    if (hb.bit > L.in_bits) { INFL_FAIL(L, 9); return; }
    infl_seek(L, hb.bit);
    L.state = INFL_ST_DECODE;
}

// ---- one trip -------------------------------------------------------------------------------------------------------------------------------------
// What a trip costs is decided by what it WAITS for.  The vector memory counter of gfx9 is one in-order counter for loads and stores: "wait until at most N of my
// memory operations are outstanding".  The compiler can only use N > 0 where it can count the operations between a load and its use - a load or store inside a
// branch that the wave may skip makes that count zero, and every wait a wait for everything, i.e. for the slowest lane's HBM miss of the newest load.  So every
// memory operation of a trip is issued in EVERY trip by EVERY lane - a lane with nothing to store stores to its own idle word of global scratch, one with nothing
// to load loads from there - and a chunk of a match is stored INFL_DEPTH trips after it was requested (the loop is unrolled INFL_DEPTH times so that the slot a
// trip works on is a constant): its load has INFL_DEPTH trips to come back, with 6 memory operations issued per trip in between.
//
// Order of the bytes.  Literals are stored at once.  A match is parked when decoded (the output position moves on at once), becomes the active match when the one
// before it has requested all its chunks, and requests 4 bytes per trip.  A request reads out[c_dst - dist, +4): the chunks requested in the last INFL_DEPTH - 1
// trips are not stored yet, so a match with a distance below 4 INFL_DEPTH requests a chunk only when no chunk is pending (one chunk per INFL_DEPTH trips), and a
// match becomes active while chunks of its predecessors are pending only if everything it reads lies in front of all of them (a pending chunk can be far behind
// the output position: a short-distance match waits for empty slots while the literals behind it are decoded and stored).
INFL_FN uint32_t infl_fix_short_distance(uint32_t w, uint32_t dist) {   // the first `dist` (1..3) bytes of w repeated over the four
    return dist == 1u ? (w & 0xffu) * 0x01010101u : (dist == 2u ? (w & 0xffffu) * 0x00010001u : (w & 0xffffffu) | (w << 24));
}
typedef uint16_t __attribute__((aligned(1), may_alias)) infl_u16u;
template <int PAR>
INFL_FN void infl_step(InflLane& L) {
    uint8_t* const idle = reinterpret_cast<uint8_t*>(L.G + (INFL_LENS_WORDS - 1) * L.g_stride);
    uint8_t* lit_at = idle; uint32_t lit = 0;
    if (L.state == INFL_ST_DECODE) {
        const uint16_t* lt = reinterpret_cast<const uint16_t*>(L.S + INFL_OFF_LT);
        const uint32_t x = infl_peek(L, L.bo);
        const uint32_t e = infl_lookup(lt, INFL_LR, x);
        const uint32_t len = e & 15u, pay = e >> 4;
        if (pay < 256u && len != 0u) {                                    // literal
            if (L.pos < L.cap) { lit_at = L.out + L.pos; lit = pay; L.pos++; L.bo += len; }
            else INFL_FAIL(L, 10);
        } else if (pay & 0x800u) {                                        // length + distance: parked behind the active match, if that place is free
            if (L.m_len == 0u) {
                const uint32_t xb = (pay >> 8) & 7u;
                const uint32_t mlen = 3u + (pay & 255u) + ((x >> len) & ((1u << xb) - 1u));
                L.bo += len + xb;
                infl_shift(L);
                const uint16_t* dt = reinterpret_cast<const uint16_t*>(L.S + INFL_OFF_DT);
                const uint32_t y = infl_peek(L, L.bo);
                const uint32_t d = infl_lookup(dt, INFL_DR, y);
                const uint32_t dl = d & 15u, ds = d >> 4;
                const uint32_t dsv = ds & 31u;                              // (a symbol that must not occur fails below; its arithmetic stays defined)
                const uint32_t dxb = dsv < 4u ? 0u : (dsv - 2u) >> 1;
                const uint32_t dist = (dsv < 4u ? 1u + dsv : 1u + ((2u + (dsv & 1u)) << dxb)) + ((y >> dl) & ((1u << dxb) - 1u));
                L.bo += dl + dxb;
                if (dl == 0u || ds >= 30u || dist > L.pos || L.pos + mlen > L.cap) INFL_FAIL(L, 11);
                else { L.m_dist = dist; L.m_dst = L.pos; L.m_len = mlen; L.pos += mlen; }
            } else { INFL_STAT(2, 1); }
        } else if (pay == 256u && len != 0u) {                            // end of block
            L.bo += len;
            L.state = L.fin ? INFL_ST_DONE : INFL_ST_HEADER;
            if (L.fin && (L.pos != L.cap || infl_bitpos(L) > L.in_bits)) INFL_FAIL(L, 12);
        } else INFL_FAIL(L, 13);                                    // no such code / a symbol that must not occur
    }
#ifndef INFL_PROBE_NOLIT                                                  /* (probe builds of tools/micro/inflate_lanes_bench.hip switch parts of a trip off) */
    *lit_at = (uint8_t)lit;
#endif
    infl_shift(L);
    L.pw = infl_word_at(L, L.in_at);
    // ---- the copy engine
#ifdef INFL_PROBE_NOCOPY
    L.c_rem = 0; L.m_len = 0;
#else
    {   // the chunk requested INFL_DEPTH trips ago
        const uint32_t sn = L.s_n[PAR], n = sn & 15u, fd = sn >> 4;
        uint32_t w = L.s_data[PAR];
        if (fd) w = infl_fix_short_distance(w, fd);
        uint8_t* const a = L.out + L.s_dst[PAR];
        *reinterpret_cast<infl_u32u*>(n == 4u ? a : idle) = w;
        const bool two = (n & 2u) != 0u && n != 4u, one = (n & 1u) != 0u;
        *reinterpret_cast<infl_u16u*>(two ? a : idle) = (uint16_t)w;
        *((one ? a + (two ? 2 : 0) : idle)) = (uint8_t)(two ? w >> 16 : w);
        L.n_pend -= n ? 1u : 0u;
        L.s_n[PAR] = 0;
    }
    // (the bytes the parked match reads in front of its own output end at m_dst - m_dist + min(m_len, m_dist): it may start beside pending chunks if all of them
    // lie behind that - p_min is the destination of the oldest chunk requested since the slots were last empty, and destinations only grow)
    if (L.c_rem == 0u && L.m_len != 0u && (L.n_pend == 0u || L.m_dst - L.m_dist + (L.m_len < L.m_dist ? L.m_len : L.m_dist) <= L.p_min)) {
        L.c_rem = L.m_len; L.c_dst = L.m_dst; L.c_dist = L.m_dist; L.m_len = 0;
    }
    {
        const bool req = L.c_rem != 0u && (L.c_dist >= 4u * INFL_DEPTH || L.n_pend == 0u);
        L.s_data[PAR] = infl_ld32u(req ? L.out + (L.c_dst - L.c_dist) : idle);
        if (req) {
            const uint32_t n = L.c_rem < 4u ? L.c_rem : 4u;
            if (L.n_pend == 0u) L.p_min = L.c_dst;
            L.s_dst[PAR] = L.c_dst; L.s_n[PAR] = n | (L.c_dist < 4u ? L.c_dist << 4 : 0u); L.n_pend++;
            L.c_dst += n; L.c_rem -= n;
        }
    }
#endif
}
INFL_FN bool infl_running(const InflLane& L) { return L.state <= INFL_ST_HEADER || L.c_rem != 0u || L.m_len != 0u || L.n_pend != 0u; }
