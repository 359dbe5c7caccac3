// inflate_core.hpp - raw DEFLATE (RFC 1951) decoder for one BGZF block, written once for two executors:
//   * the GPU: one 64-lane wavefront per block (bgzf.hip).  Huffman decoding is inherently serial, so the decode state (bit buffer, canonical
//     code counts) is WAVE-UNIFORM - the compiler keeps it in scalar registers and the scalar unit does the decoding - while the vector lanes do
//     what is parallel: literals are staged one per lane and leave as 64-byte stores, matches and stored blocks are copied 64 bytes per step.
//   * the host (INF_HOST, tests/test_inflate_core.py through tools/inflate_host_test.cpp): the same algorithm with the lane operations
//     emulated, checked against zlib on real BGZF blocks.
// Replaces: zlib's inflate() as htslib / pysam use it under pysam.AlignmentFile (the reference reads BAM through them, SVIM_COLLECT.py:132-137).
// Canonical decoding after Mark Adler's description of the format (count of codes per length + symbols in code order); nothing else is shared.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define INF_MAXBITS 15
#define INF_MAXL 288
#define INF_MAXD 32

#define INF_E_BLOCKTYPE (-1)
#define INF_E_STORED (-2)
#define INF_E_CODELEN (-3)
#define INF_E_OVERSUB (-4)
#define INF_E_SYMBOL (-5)
#define INF_E_DIST (-6)
#define INF_E_OUTPUT (-7)
#define INF_E_INPUT (-8)

// per-wave scratch (LDS on the device): code lengths while a dynamic header is read, then the symbols of both codes in canonical order
#define INF_FAST_L 10          /* literal/length codes of up to 10 bits and distance codes of up to 8 bits are decoded by ONE table lookup; */
#define INF_FAST_D 8           /* an entry is (code length << 9) | symbol, 0 = longer code: canonical walk                              */
struct InfScratch {
    uint16_t len[INF_MAXL + INF_MAXD];
    uint16_t lsym[INF_MAXL];
    uint16_t dsym[INF_MAXD];
    uint16_t cnt[INF_MAXBITS + 1];
    uint16_t code[INF_MAXL];                 // canonical code of every symbol (while a fast table is filled)
    uint16_t fast_l[1 << INF_FAST_L];
    uint16_t fast_d[1 << INF_FAST_D];
};

#ifdef INF_HOST
#define INF_FN static inline
#define INF_LANE 0
#define INF_UNI(x) (x)
#else
#define INF_FN __device__ __forceinline__
#define INF_LANE ((int)(threadIdx.x & 63))
#define INF_UNI(x) __builtin_amdgcn_readfirstlane((int)(x))
#endif

struct InfState {
    // input: 32-bit words (the block's payload starts at a 4-byte aligned address), bit buffer
    const uint32_t* in; uint32_t in_words, in_at;      // in_at: words handed to the bit buffer so far
    uint32_t ahead;                                    // word in_at, loaded when word in_at - 1 was taken: its latency hides behind the symbols in between
    uint64_t bitbuf; int bitcnt;
    // output
    uint8_t* out; uint32_t out_cap, pos;      // pos: bytes produced (staged ones included)
    uint32_t staged;                           // literals waiting in the lanes (0..64); they belong to out[pos - staged, pos)
    uint32_t clean;                            // every store to out[0, clean) is known to have completed (the last wait)
#ifdef INF_HOST
    uint8_t stage[64];
#else
    uint32_t stage;                            // this lane's staged byte
#endif
};

INF_FN void inf_need(InfState& s, int n) {                 // n <= 32
    if (s.bitcnt < n) {
        const uint32_t w = (uint32_t)INF_UNI(s.ahead);
        s.in_at++;
        s.ahead = s.in_at < s.in_words ? s.in[s.in_at] : 0u;       // reading past the end yields zeros; the caller notices by position
        s.bitbuf |= (uint64_t)w << s.bitcnt;
        s.bitcnt += 32;
    }
}
INF_FN uint32_t inf_bits(InfState& s, int n) {             // n <= 24
    inf_need(s, n);
    const uint32_t v = (uint32_t)s.bitbuf & ((1u << n) - 1u);
    s.bitbuf >>= n; s.bitcnt -= n;
    return v;
}

// ---- output -----------------------------------------------------------------------------------------------------------------
INF_FN void inf_flush(InfState& s) {
    if (!s.staged) return;
    const uint32_t base = s.pos - s.staged;
#ifdef INF_HOST
    for (uint32_t i = 0; i < s.staged; i++) s.out[base + i] = s.stage[i];
#else
    if ((uint32_t)INF_LANE < s.staged) s.out[base + INF_LANE] = (uint8_t)s.stage;
#endif
    s.staged = 0;
}
INF_FN int inf_literal(InfState& s, uint32_t byte) {
    if (s.pos >= s.out_cap) return INF_E_OUTPUT;
#ifdef INF_HOST
    s.stage[s.staged] = (uint8_t)byte;
#else
    if ((uint32_t)INF_LANE == s.staged) s.stage = byte;
#endif
    s.staged++; s.pos++;
    if (s.staged == 64) inf_flush(s);
    return 0;
}
// out[pos, pos + len) = out[pos - dist ...] with the overlap rule of LZ77 (a distance shorter than the length repeats the pattern)
INF_FN int inf_match(InfState& s, uint32_t dist, uint32_t len) {
    if (dist > s.pos) return INF_E_DIST;
    if (s.pos + len > s.out_cap) return INF_E_OUTPUT;
    inf_flush(s);
    // the source may have been written by this wave a moment ago (by other lanes): only then wait for the stores (workgroup-scope fence)
    if (s.pos - dist + (len < dist ? len : dist) > s.clean) {
#ifndef INF_HOST
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
        s.clean = s.pos;
    }
#ifdef INF_HOST
    for (uint32_t i = 0; i < len; i++) s.out[s.pos + i] = s.out[s.pos - dist + i];
#else
    const uint8_t* src = s.out + (s.pos - dist);
    uint8_t* dst = s.out + s.pos;
    for (uint32_t i = (uint32_t)INF_LANE; i < len; i += 64) dst[i] = src[dist >= len ? i : i % dist];
#endif
    s.pos += len;
    return 0;
}

// ---- canonical Huffman ------------------------------------------------------------------------------------------------------
// counts per code length live in registers (constant indices after unrolling); the symbols in canonical order in scratch
struct InfCounts { uint16_t c[INF_MAXBITS + 1]; };

INF_FN int inf_decode(InfState& s, const InfCounts& h, const uint16_t* symbol) {
    inf_need(s, INF_MAXBITS);
    uint32_t bits = (uint32_t)s.bitbuf;
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= INF_MAXBITS; len++) {
        code |= (int)(bits & 1u); bits >>= 1;
        const int count = h.c[len];
        if (code - count < first) {
            s.bitbuf >>= len; s.bitcnt -= len;
            return (int)INF_UNI(symbol[index + (code - first)]);
        }
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return INF_E_SYMBOL;
}

// lengths len[0..n) -> counts (registers) + symbols in canonical order; returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed
INF_FN int inf_construct(InfScratch& sc, const uint16_t* len, int n, InfCounts& h, uint16_t* symbol) {
    const bool writer = INF_LANE == 0;
    if (writer) for (int l = 0; l <= INF_MAXBITS; l++) sc.cnt[l] = 0;
    if (writer) for (int i = 0; i < n; i++) sc.cnt[len[i]]++;
#pragma unroll
    for (int l = 0; l <= INF_MAXBITS; l++) h.c[l] = (uint16_t)INF_UNI(sc.cnt[l]);
    if (h.c[0] == n) return 0;                                 // no codes: complete, but decoding with it fails
    int left = 1;
#pragma unroll
    for (int l = 1; l <= INF_MAXBITS; l++) { left <<= 1; left -= h.c[l]; if (left < 0) return left; }
    // offsets of each length in the symbol table (sc.cnt is reused as the running offsets)
    if (writer) {
        uint16_t off = 0;
        for (int l = 1; l <= INF_MAXBITS; l++) { const uint16_t c = sc.cnt[l]; sc.cnt[l] = off; off = (uint16_t)(off + c); }
        for (int i = 0; i < n; i++) if (len[i] != 0) symbol[sc.cnt[len[i]]++] = (uint16_t)i;
    }
    return left;
}

// fast table of a code built by inf_construct: entry[bits] for every `fb`-bit window whose low bits are a complete code of at most fb bits
INF_FN void inf_fast_table(InfScratch& sc, const uint16_t* len, int n, const InfCounts& h, uint16_t* fast, int fb) {
    // canonical codes: first code of every length, then symbols in order (lane 0; a few hundred steps)
    if (INF_LANE == 0) {
        uint16_t* next = sc.cnt;                                    // free again after inf_construct
        uint32_t code = 0;
        next[0] = 0;
#pragma unroll
        for (int l = 1; l <= INF_MAXBITS; l++) { code = (code + (l > 1 ? (uint32_t)h.c[l - 1] : 0u)) << 1; next[l] = (uint16_t)code; }
        for (int i = 0; i < n; i++) { const int l = len[i]; sc.code[i] = l ? next[l]++ : 0; }
    }
#ifdef INF_HOST
    for (int k = 0; k < (1 << fb); k++) fast[k] = 0;
    for (int i = 0; i < n; i++) {
#else
    for (int k = INF_LANE; k < (1 << fb); k += 64) fast[k] = 0;
    for (int i = INF_LANE; i < n; i += 64) {
#endif
        const int l = len[i];
        if (l == 0 || l > fb) continue;
        uint32_t c = sc.code[i], r = 0;
        for (int b = 0; b < l; b++) { r = (r << 1) | (c & 1u); c >>= 1; }          // the stream carries codes most significant bit first
        const uint16_t e = (uint16_t)((l << 9) | i);
        for (uint32_t k = r; k < (1u << fb); k += 1u << l) fast[k] = e;
    }
}

INF_FN int inf_decode_fast(InfState& s, const InfCounts& h, const uint16_t* symbol, const uint16_t* fast, int fb) {
    inf_need(s, INF_MAXBITS);
    const uint32_t e = (uint32_t)INF_UNI(fast[(uint32_t)s.bitbuf & ((1u << fb) - 1u)]) & 0xffffu;
    if (e) { const int l = (int)(e >> 9); s.bitbuf >>= l; s.bitcnt -= l; return (int)(e & 511u); }
    return inf_decode(s, h, symbol);
}

INF_FN int inf_codes(InfState& s, const InfCounts& lc, const uint16_t* lsym, const InfCounts& dc, const uint16_t* dsym, const InfScratch& sc) {
    // length / distance bases and extra bits (RFC 1951 3.2.5), packed: base | extra << 16
    static const uint32_t lens[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11 | 1 << 16, 13 | 1 << 16, 15 | 1 << 16, 17 | 1 << 16, 19 | 2 << 16, 23 | 2 << 16, 27 | 2 << 16,
                                      31 | 2 << 16, 35 | 3 << 16, 43 | 3 << 16, 51 | 3 << 16, 59 | 3 << 16, 67 | 4 << 16, 83 | 4 << 16, 99 | 4 << 16, 115 | 4 << 16,
                                      131 | 5 << 16, 163 | 5 << 16, 195 | 5 << 16, 227 | 5 << 16, 258};
    static const uint32_t dists[30] = {1, 2, 3, 4, 5 | 1 << 16, 7 | 1 << 16, 9 | 2 << 16, 13 | 2 << 16, 17 | 3 << 16, 25 | 3 << 16, 33 | 4 << 16, 49 | 4 << 16,
                                       65 | 5 << 16, 97 | 5 << 16, 129 | 6 << 16, 193 | 6 << 16, 257 | 7 << 16, 385 | 7 << 16, 513 | 8 << 16, 769 | 8 << 16,
                                       1025 | 9 << 16, 1537 | 9 << 16, 2049 | 10 << 16, 3073 | 10 << 16, 4097 | 11 << 16, 6145 | 11 << 16, 8193 | 12 << 16,
                                       12289 | 12 << 16, 16385 | 13 << 16, 24577 | 13 << 16};
    for (;;) {
        int sym = inf_decode_fast(s, lc, lsym, sc.fast_l, INF_FAST_L);
        if (sym < 0) return sym;
        if (sym < 256) { const int rc = inf_literal(s, (uint32_t)sym); if (rc) return rc; continue; }
        if (sym == 256) return 0;
        sym -= 257;
        if (sym >= 29) return INF_E_SYMBOL;
        const uint32_t le = lens[sym];
        const uint32_t len = (le & 0xffffu) + inf_bits(s, (int)(le >> 16));
        const int ds = inf_decode_fast(s, dc, dsym, sc.fast_d, INF_FAST_D);
        if (ds < 0) return ds;
        if (ds >= 30) return INF_E_SYMBOL;
        const uint32_t de = dists[ds];
        const uint32_t dist = (de & 0xffffu) + inf_bits(s, (int)(de >> 16));
        const int rc = inf_match(s, dist, len);
        if (rc) return rc;
    }
}

// one raw DEFLATE stream: `payload` (any alignment, in_bytes long), out_cap = ISIZE.  Returns the number of bytes produced (== ISIZE for a
// sound block) or a negative INF_E_*.  Device: every lane of the wave must call, with wave-uniform arguments.
INF_FN int inflate_raw(const uint8_t* payload, uint32_t in_bytes, uint8_t* out, uint32_t out_cap, InfScratch& sc) {
    InfState s;
    // the bit reader works on aligned 32-bit words: the up to 3 bytes in front of the payload are read and dropped
    const uint32_t skip = (uint32_t)(reinterpret_cast<uintptr_t>(payload) & 3u);
    in_bytes += skip;
    s.in = reinterpret_cast<const uint32_t*>(payload - skip); s.in_words = (in_bytes + 3u) / 4u; s.in_at = 0; s.bitbuf = 0; s.bitcnt = 0;
    s.ahead = s.in_words ? s.in[0] : 0u;
    s.out = out; s.out_cap = out_cap; s.pos = 0; s.staged = 0; s.clean = 0;
#ifndef INF_HOST
    s.stage = 0;
#endif
    if (skip) (void)inf_bits(s, 8 * (int)skip);
    InfCounts lc, dc;
    for (;;) {
        const uint32_t last = inf_bits(s, 1), type = inf_bits(s, 2);
        if (type == 0) {
            // stored: skip to the byte boundary, LEN, ~LEN, LEN bytes
            const int drop = s.bitcnt & 7;
            s.bitbuf >>= drop; s.bitcnt -= drop;
            const uint32_t len = inf_bits(s, 16), nlen = inf_bits(s, 16);
            if (len != (~nlen & 0xffffu)) return INF_E_STORED;
            if (s.pos + len > s.out_cap) return INF_E_OUTPUT;
            inf_flush(s);
            // byte position of the data in the input: words consumed * 4 - bytes still in the bit buffer
            const uint32_t at = s.in_at * 4u - (uint32_t)(s.bitcnt >> 3);
            if (at + len > in_bytes) return INF_E_INPUT;
            const uint8_t* src = reinterpret_cast<const uint8_t*>(s.in) + at;
#ifdef INF_HOST
            for (uint32_t i = 0; i < len; i++) s.out[s.pos + i] = src[i];
#else
            for (uint32_t i = (uint32_t)INF_LANE; i < len; i += 64) s.out[s.pos + i] = src[i];
#endif
            s.pos += len;
            // restart the bit reader behind the stored bytes
            const uint32_t next = at + len;
            s.in_at = next / 4u; s.bitbuf = 0; s.bitcnt = 0;
            s.ahead = s.in_at < s.in_words ? s.in[s.in_at] : 0u;
            if (next & 3u) (void)inf_bits(s, 8 * (int)(next & 3u));
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                if (INF_LANE == 0) {
                    for (int i = 0; i < 144; i++) sc.len[i] = 8;
                    for (int i = 144; i < 256; i++) sc.len[i] = 9;
                    for (int i = 256; i < 280; i++) sc.len[i] = 7;
                    for (int i = 280; i < 288; i++) sc.len[i] = 8;
                    for (int i = 0; i < 30; i++) sc.len[288 + i] = 5;
                }
                (void)inf_construct(sc, sc.len, 288, lc, sc.lsym);
                inf_fast_table(sc, sc.len, 288, lc, sc.fast_l, INF_FAST_L);
                (void)inf_construct(sc, sc.len + 288, 30, dc, sc.dsym);
                inf_fast_table(sc, sc.len + 288, 30, dc, sc.fast_d, INF_FAST_D);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)inf_bits(s, 5) + 257, ndist = (int)inf_bits(s, 5) + 1, ncode = (int)inf_bits(s, 4) + 4;
                if (nlen > 286 || ndist > 30) return INF_E_CODELEN;
                // the code-length code: 19 symbols of up to 7 bits; its lengths go through sc.len as well (lane 0 writes, everybody reads)
                for (int i = 0; i < 19; i++) { const uint32_t v = i < ncode ? inf_bits(s, 3) : 0u; if (INF_LANE == 0) sc.len[order[i]] = (uint16_t)v; }
                InfCounts cc;
                int err = inf_construct(sc, sc.len, 19, cc, sc.lsym);
                if (err != 0) return INF_E_CODELEN;                      // complete code required here
                // cc's symbols sit in sc.lsym[0..19); move them out of the way of the real table: the tail of dsym's neighbour is free until then
                uint16_t* csym = sc.lsym + 256;                           // lsym[256..275): overwritten only when the literal/length table is built
                if (INF_LANE == 0) for (int i = 0; i < 19; i++) csym[i] = sc.lsym[i];
                int index = 0;
                while (index < nlen + ndist) {
                    int sym = inf_decode(s, cc, csym);
                    if (sym < 0) return sym;
                    if (sym < 16) { if (INF_LANE == 0) sc.len[index] = (uint16_t)sym; index++; }
                    else {
                        uint32_t prev = 0, rep;
                        if (sym == 16) { if (index == 0) return INF_E_CODELEN; prev = (uint32_t)INF_UNI(sc.len[index - 1]); rep = 3 + inf_bits(s, 2); }
                        else if (sym == 17) rep = 3 + inf_bits(s, 3);
                        else rep = 11 + inf_bits(s, 7);
                        if (index + (int)rep > nlen + ndist) return INF_E_CODELEN;
                        if (INF_LANE == 0) for (uint32_t r = 0; r < rep; r++) sc.len[index + (int)r] = (uint16_t)prev;
                        index += (int)rep;
                    }
                }
                if ((int)INF_UNI(sc.len[256]) == 0) return INF_E_CODELEN;           // no end-of-block code
                // the distance lengths first (construct of the literal/length code overwrites lsym, where csym lived)
                err = inf_construct(sc, sc.len + nlen, ndist, dc, sc.dsym);
                if (err < 0 || (err > 0 && ndist - dc.c[0] != 1)) return INF_E_OVERSUB;   // incomplete only allowed for a single distance code
                inf_fast_table(sc, sc.len + nlen, ndist, dc, sc.fast_d, INF_FAST_D);
                err = inf_construct(sc, sc.len, nlen, lc, sc.lsym);
                if (err < 0 || (err > 0 && nlen - lc.c[0] != 1)) return INF_E_OVERSUB;
                inf_fast_table(sc, sc.len, nlen, lc, sc.fast_l, INF_FAST_L);
            }
            const int rc = inf_codes(s, lc, sc.lsym, dc, sc.dsym, sc);
            if (rc) return rc;
        } else return INF_E_BLOCKTYPE;
        if (last) break;
    }
    inf_flush(s);
    return (int)s.pos;
}
