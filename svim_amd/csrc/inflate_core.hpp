// inflate_core.hpp - raw DEFLATE (RFC 1951) decoder for one BGZF block, written for a 64-lane wavefront (one wave per block, bgzf.hip) and, with the
// lane operations emulated, for the host (INF_HOST: tests/test_inflate_core.py through tools/inflate_host_test.cpp, fuzzed against zlib).
// Replaces: zlib's inflate() as htslib / pysam use it under pysam.AlignmentFile (the reference reads BAM through them, SVIM_COLLECT.py:132-137).
//
// Huffman decoding is a serial chain (a code's length decides where the next one starts), but only THAT is serial.  Per step the wave
//   1. looks up, in ONE LDS gather, the table entry of the code that would start at each of the next 64 bit offsets (lane l: offset l);
//   2. follows the chain of real code starts through those 64 entries with v_readlane - a few scalar instructions per symbol and no memory
//      round trip - as long as the symbols are literals, collecting their offsets in a 64-bit mask;
//   3. writes all literals of the step with one predicated LDS byte store (lane = code start, position = rank of its bit in the mask).
// A length symbol ends the run: extra bits and the distance code are peeked from registers (the compressed input lives in two VGPRs, 256 bytes
// each, refilled 2048 bits ahead), and the match is copied inside an LDS RING holding the most recent output (4 KiB: what BAM matches mostly
// reach for); only a source farther back is read from global memory, behind a fence.  The ring is the write buffer as well: output leaves as
// aligned 16-byte-per-lane stores, 1 KiB per instruction.  Code tables are built by all lanes (ballot ranks instead of a serial counting pass).
// Canonical decoding after Mark Adler's description of the format (count of codes per length + symbols in code order); nothing else is shared.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>

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

#ifndef INF_FAST_L
#define INF_FAST_L 10          /* literal/length codes of up to 10 bits, distance codes of up to 8 and code-length codes of up to 7 are decoded by ONE lookup */
#endif
#define INF_FAST_D 8
#define INF_FAST_C 7
#ifndef INF_RING
#define INF_RING 4096u         /* bytes of recent output kept in LDS (power of two, multiple of 16) */
#endif
#define INF_RMASK (INF_RING - 1u)
#define INF_FLUSH_AT 1039u     /* pending bytes that trigger a flush: 1 KiB + what alignment may hold back */
/* table entry: code length << 10 | flag 0x200 ("simple": a literal / a plain code length) | symbol; 0 = code longer than the table: canonical walk */
#define INF_SIMPLE 0x200u

// per-wave scratch (LDS on the device)
struct InfScratch {
    alignas(16) uint8_t ring[INF_RING];
    uint16_t fast_l[1 << INF_FAST_L];
    uint16_t fast_d[1 << INF_FAST_D];
    union {
        uint16_t fast_c[1 << INF_FAST_C];       // code-length code: while a header is read
        uint32_t pk_list[64];                   // ... afterwards: the tokens of a multi-window step in output order (inf_emit_bytes_multi)
    };
    uint16_t lsym[INF_MAXL];                 // symbols in canonical order (codes longer than the fast tables)
    uint16_t dsym[INF_MAXD];
    uint16_t csym[20];
    uint16_t cnt_l[16], cnt_d[16], cnt_c[16];   // codes per length (canonical walk of the long codes)
    union {
        uint8_t len[INF_MAXL + INF_MAXD + 16];  // code lengths while a header is read
        struct {                                // ... and, once the tables are built, what the token decode of the block needs:
            uint32_t dist_lut[32];              // distance symbol -> base | extra bits << 16 (0: not a symbol)
            uint16_t len_lut[32];               // length symbol - 257 -> base | extra bits << 9 (0: not a symbol)
            uint8_t tokmap[64];                 // output byte of a step -> 1 + lane (rank, in a multi-window step) of the token that starts there, 0 elsewhere
        };
    };
};

// ---- lane abstraction: the device executes a "vector" statement in every lane; the host runs the 64 lanes in a loop ------------------------------
#ifdef INF_HOST
#define INF_FN static inline
#define W_VEC(T, name) T name[64]
#define W_VEC2(T, name, k) T name[k][64]
#define W_FOR for (int lane_ = 0; lane_ < 64; lane_++)
#define V(name) name[lane_]
#define V2(name, r) name[r][lane_]
#define V2W(name, r) name[r]
#define W_LANE lane_
#define W_READLANE(name, idx) ((uint32_t)name[(idx)])
#define W_BALLOT(dst, expr) do { dst = 0; for (int lane_ = 0; lane_ < 64; lane_++) if (expr) dst |= 1ull << lane_; } while (0)
#define W_RANK(mask) ((uint32_t)__builtin_popcountll((mask) & ((1ull << lane_) - 1ull)))
#define W_FENCE() do { } while (0)
#define W_WAIT_LOADS() do { } while (0)
static inline uint32_t inf_bitrev(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; i++) { r = (r << 1) | (x & 1u); x >>= 1; } return r; }
#else
#define INF_FN __device__ __forceinline__
#define W_VEC(T, name) T name
#define W_VEC2(T, name, k) T name[k]
#define W_FOR
#define V(name) name
#define V2(name, r) name[r]
#define V2W(name, r) name[r]
#define W_LANE ((int)(threadIdx.x & 63))
#define W_READLANE(name, idx) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(idx)))
#define W_BALLOT(dst, expr) dst = __ballot(expr)
#define W_RANK(mask) ((uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)((mask) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(mask), 0u)))
#define W_WAIT_LOADS() __builtin_amdgcn_s_waitcnt(0x0F70)      /* s_waitcnt vmcnt(0) */
#define W_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
static __device__ __forceinline__ uint32_t inf_bitrev(uint32_t x) { return __builtin_bitreverse32(x); }
#endif


// exclusive prefix sum of a per-lane value over the 64 lanes (dst) and the sum of all lanes (total, wave-uniform)
#ifdef INF_HOST
#define W_EXCL_SCAN(dst, src, total) do { uint32_t run_ = 0; for (int lane_ = 0; lane_ < 64; lane_++) { const uint32_t v_ = src[lane_]; dst[lane_] = run_; run_ += v_; } total = run_; } while (0)
#else
#define INF_DPP_ADD(d_, s_, ctrl_, row_, bank_) d_ += __builtin_amdgcn_update_dpp(0, s_, ctrl_, row_, bank_, true)
#define W_EXCL_SCAN(dst, src, total) do { int i_ = (int)(src), v_ = i_;                                                  \
        INF_DPP_ADD(i_, v_, 0x111, 0xf, 0xf); INF_DPP_ADD(i_, v_, 0x112, 0xf, 0xf); INF_DPP_ADD(i_, v_, 0x113, 0xf, 0xf);   \
        INF_DPP_ADD(i_, i_, 0x114, 0xf, 0xe); INF_DPP_ADD(i_, i_, 0x118, 0xf, 0xc);                                         \
        INF_DPP_ADD(i_, i_, 0x142, 0xa, 0xf); INF_DPP_ADD(i_, i_, 0x143, 0xc, 0xf);                                         \
        dst = (uint32_t)(i_ - v_); total = (uint32_t)__builtin_amdgcn_readlane(i_, 63); } while (0)
#endif
// dst[lane] = src[idx[lane]] (idx < 64), and an inclusive running maximum over the lanes (values >= 0)
#ifdef INF_HOST
#define W_BPERMUTE(dst, src, idx) do { uint32_t t_[64]; for (int l_ = 0; l_ < 64; l_++) t_[l_] = src[(idx[l_]) & 63u]; for (int l_ = 0; l_ < 64; l_++) dst[l_] = t_[l_]; } while (0)
#define W_INCL_MAX_SCAN(v) do { uint32_t run_ = 0; for (int l_ = 0; l_ < 64; l_++) { if (v[l_] > run_) run_ = v[l_]; v[l_] = run_; } } while (0)
#else
#define W_BPERMUTE(dst, src, idx) dst = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx) << 2), (int)(src))
#define INF_DPP_MAX(d_, s_, ctrl_, row_, bank_) { const int t_ = __builtin_amdgcn_update_dpp(0, s_, ctrl_, row_, bank_, true); d_ = t_ > d_ ? t_ : d_; }
#define W_INCL_MAX_SCAN(v) do { int i_ = (int)(v), v_ = i_;                                                            \
        INF_DPP_MAX(i_, v_, 0x111, 0xf, 0xf) INF_DPP_MAX(i_, v_, 0x112, 0xf, 0xf) INF_DPP_MAX(i_, v_, 0x113, 0xf, 0xf)   \
        { const int u_ = i_; INF_DPP_MAX(i_, u_, 0x114, 0xf, 0xe) } { const int u_ = i_; INF_DPP_MAX(i_, u_, 0x118, 0xf, 0xc) }  \
        { const int u_ = i_; INF_DPP_MAX(i_, u_, 0x142, 0xa, 0xf) } { const int u_ = i_; INF_DPP_MAX(i_, u_, 0x143, 0xc, 0xf) }  \
        v = (uint32_t)i_; } while (0)
#endif
/* a token (literal, or length + distance) that would start at a bit offset: bits it takes | output bytes << 8 | flags */
#define INF_T_MATCH (1u << 20)
#define INF_T_STOP  (1u << 21)
#define INF_T_FAR   (1u << 22)   /* a match whose source is no longer (safely) in the ring: read back from the flushed output */
#ifndef INF_STEP_CAP
#define INF_STEP_CAP 1024u
#endif
/* INF_STEP_CAP: output bytes one step may produce: its literals are written before its matches are copied, so a match of the step must not
                                  reach back farther than ring size - this (a later literal of the same step would already sit on its source) */

struct InfState {
    // input: aligned 32-bit words, 256 of them in four registers: lane l of c[r] holds word cbase + 4 l + r; bp = bit position from the aligned base
    const uint32_t* in; uint32_t in_words, cbase, bp;
    W_VEC2(uint32_t, c, 4);
    // output
    uint8_t* out; uint32_t out_cap, pos, flushed, clean, mis;      // mis: (address of out) & 15 - ring index of output byte p is (p + mis) & INF_RMASK
    InfScratch* sc;
#ifdef INF_PROFILE
    uint32_t prof[16]; uint64_t t_last;
#endif
};

// optional cycle accounting of the decode loop (build with -DINF_PROFILE; tools/inflate_profile.py reads the totals): INF_PROF charges the cycles since the
// previous mark to a slot, INF_COUNT adds to a counter
#if defined(INF_PROFILE) && !defined(INF_HOST)
__device__ unsigned long long g_inf_prof[16];
#define INF_PROF(s, slot) { const uint64_t now_ = __builtin_readcyclecounter(); (s).prof[slot] += (uint32_t)(now_ - (s).t_last); (s).t_last = now_; }
#define INF_COUNT(s, slot, n) { (s).prof[slot] += (uint32_t)(n); }
#else
#define INF_PROF(s, slot)
#define INF_COUNT(s, slot, n)
#endif
#ifndef INF_STAT
#define INF_STAT(path, len, dist)       // host-side token statistics (tools/inflate_host_test.cpp --stats)
#endif
#ifndef INF_STEP_STAT
#define INF_STEP_STAT()                 // host-side: one decode step begins
#endif
#ifndef INF_WIDE_STAT
#define INF_WIDE_STAT(taken)            // host-side: a multi-window step was tried and taken with `taken` windows / given up for an ordinary one (0)
#endif

// ---- input ----------------------------------------------------------------------------------------------------------------------------------------
// 1 KiB of input per refill, 16 bytes per lane, loaded when a step starts in the last words of the buffer.  No prefetch register: a register with a load
// in flight makes the compiler wait (for every memory operation of the wave) wherever that register could be read, i.e. at every step; one exposed round
// trip per KiB costs less, also when the input sits behind PCIe
#define INF_IN_WORDS 256u
#define INF_IN_SLACK 16u           /* words a step may read beyond its first one */
INF_FN void inf_load_chunks(InfState& s, uint32_t cbase) {
    s.cbase = cbase;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        W_FOR { const uint32_t k = cbase + 4u * (uint32_t)W_LANE + (uint32_t)r; V2(s.c, r) = k < s.in_words ? s.in[k] : 0u; }
    }
    W_WAIT_LOADS();                    // here, once per KiB - otherwise every step has to assume that the four registers may still be on their way
}
// called at the top of every decode step: the words the step reads (bp >> 5 and INF_IN_SLACK more) are in the buffer
INF_FN void inf_sync_input(InfState& s) {
    if ((s.bp >> 5) - s.cbase > INF_IN_WORDS - INF_IN_SLACK) inf_load_chunks(s, (s.bp >> 5) & ~3u);
}
INF_FN uint32_t inf_word(const InfState& s, uint32_t d) {             // word d, cbase <= d < cbase + 256
    const uint32_t j = d - s.cbase, q = (j >> 2) & 63u, r = j & 3u;
    const uint32_t a = W_READLANE(s.c[0], q), b = W_READLANE(s.c[1], q), c = W_READLANE(s.c[2], q), e = W_READLANE(s.c[3], q);       // all four, then selects: no branch
    return r == 0u ? a : (r == 1u ? b : (r == 2u ? c : e));
}
// words d .. d + 3 (cbase <= d, d + 3 < cbase + 256): four lane reads, which registers they come from depends on d & 3 only (cbase is a multiple of 4)
#define INF_WORDS4(s, d, w0, w1, w2, w3)                                                                                                   \
    {                                                                                                                                      \
        const uint32_t j_ = (d) - (s).cbase, q_ = j_ >> 2;                                                                                 \
        switch (j_ & 3u) {                                                                                                                 \
        case 0u: w0 = W_READLANE((s).c[0], q_); w1 = W_READLANE((s).c[1], q_); w2 = W_READLANE((s).c[2], q_); w3 = W_READLANE((s).c[3], q_); break;            \
        case 1u: w0 = W_READLANE((s).c[1], q_); w1 = W_READLANE((s).c[2], q_); w2 = W_READLANE((s).c[3], q_); w3 = W_READLANE((s).c[0], q_ + 1u); break;       \
        case 2u: w0 = W_READLANE((s).c[2], q_); w1 = W_READLANE((s).c[3], q_); w2 = W_READLANE((s).c[0], q_ + 1u); w3 = W_READLANE((s).c[1], q_ + 1u); break;  \
        default: w0 = W_READLANE((s).c[3], q_); w1 = W_READLANE((s).c[0], q_ + 1u); w2 = W_READLANE((s).c[1], q_ + 1u); w3 = W_READLANE((s).c[2], q_ + 1u); break; \
        }                                                                                                                                  \
    }
INF_FN uint32_t inf_peek(const InfState& s, uint32_t bp) {            // the 32 bits from bit position bp on
    const uint32_t d = bp >> 5, sh = bp & 31u;
    const uint64_t two = (uint64_t)inf_word(s, d) | ((uint64_t)inf_word(s, d + 1u) << 32);
    return (uint32_t)(two >> sh);
}
INF_FN uint32_t inf_bits(InfState& s, int n) {                        // n <= 24
    const uint32_t v = inf_peek(s, s.bp) & ((1u << n) - 1u);
    s.bp += (uint32_t)n;
    return v;
}

// ---- output ---------------------------------------------------------------------------------------------------------------------------------------
INF_FN uint32_t inf_ridx(const InfState& s, uint32_t p) { return (p + s.mis) & INF_RMASK; }

// ring -> global for output bytes [flushed, upto): bytes up to the first 16-byte boundary of the destination, then 16 bytes per lane
INF_FN void inf_flush(InfState& s, uint32_t upto) {
    while (s.flushed < upto) {
        const uint32_t a = (s.flushed + s.mis) & 15u, left = upto - s.flushed;
        if (a != 0u || left < 16u) {
            uint32_t n = a ? 16u - a : left;
            if (n > left) n = left;
            W_VEC(uint32_t, v);
            W_FOR { if ((uint32_t)W_LANE < n) V(v) = s.sc->ring[inf_ridx(s, s.flushed + (uint32_t)W_LANE)]; }
            W_FOR { if ((uint32_t)W_LANE < n) s.out[s.flushed + (uint32_t)W_LANE] = (uint8_t)V(v); }
            s.flushed += n;
        } else {
            uint32_t n16 = left >> 4;
            if (n16 > 64u) n16 = 64u;
#ifdef INF_HOST
            for (uint32_t i = 0; i < n16; i++) memcpy(s.out + s.flushed + 16u * i, s.sc->ring + inf_ridx(s, s.flushed + 16u * i), 16);
#else
            if ((uint32_t)W_LANE < n16) {
                const uint4 q = *reinterpret_cast<const uint4*>(s.sc->ring + inf_ridx(s, s.flushed + 16u * (uint32_t)W_LANE));
                *reinterpret_cast<uint4*>(s.out + s.flushed + 16u * (uint32_t)W_LANE) = q;
            }
#endif
            s.flushed += 16u * n16;
        }
    }
}
INF_FN void inf_maybe_flush(InfState& s) {
    if (s.pos - s.flushed >= INF_FLUSH_AT) inf_flush(s, s.pos - ((s.pos + s.mis) & 15u));       // up to the last 16-byte boundary of the destination
}

// ring[op, op + len) = ring[op - dist ...] with the overlap rule of LZ77 (a distance shorter than the length repeats the pattern); dist + len <= INF_RING.
// Chunks of `step` bytes never read what they write: step = 64 for dist >= 64, else the largest multiple of dist below 65 (the first chunk lays
// the pattern out from the source, the others copy from one step back)
INF_FN void inf_copy_near(InfState& s, uint32_t op, uint32_t dist, uint32_t len) {
    uint8_t* ring = s.sc->ring;
    W_VEC(uint32_t, v);
    if (len <= 64u && dist >= len) {                                      // the common case (CIGAR words, tags): one gather, one scatter
        W_FOR { if ((uint32_t)W_LANE < len) V(v) = ring[inf_ridx(s, op + (uint32_t)W_LANE - dist)]; }
        W_FOR { if ((uint32_t)W_LANE < len) ring[inf_ridx(s, op + (uint32_t)W_LANE)] = (uint8_t)V(v); }
        return;
    }
    if (dist >= 64u || dist >= len) {
        for (uint32_t done = 0; done < len; done += 64u) {
            const uint32_t n = len - done < 64u ? len - done : 64u;
            W_FOR { if ((uint32_t)W_LANE < n) V(v) = ring[inf_ridx(s, op + done + (uint32_t)W_LANE - dist)]; }
            W_FOR { if ((uint32_t)W_LANE < n) ring[inf_ridx(s, op + done + (uint32_t)W_LANE)] = (uint8_t)V(v); }
        }
    } else {
        const uint32_t step = dist * (64u / dist);
        uint32_t n = len < step ? len : step;
        W_FOR { if ((uint32_t)W_LANE < n) V(v) = ring[inf_ridx(s, op - dist + (dist == 1u ? 0u : (uint32_t)W_LANE % dist))]; }
        W_FOR { if ((uint32_t)W_LANE < n) ring[inf_ridx(s, op + (uint32_t)W_LANE)] = (uint8_t)V(v); }
        // a step is a whole number of periods: every later chunk repeats, lane by lane, the bytes of the first one - writes only
        for (uint32_t done = n; done < len; done += step) {
            n = len - done < step ? len - done : step;
            W_FOR { if ((uint32_t)W_LANE < n) ring[inf_ridx(s, op + done + (uint32_t)W_LANE)] = (uint8_t)V(v); }
        }
    }
}

// out[pos, pos + len) = out[pos - dist ...]: inside the ring, or - a source that left the ring - from global memory
// ring[op, op + len) from the flushed output: dist + len > INF_RING - INF_STEP_CAP, so the source ends in front of everything still pending (at most
// INF_FLUSH_AT + INF_STEP_CAP bytes); the stores of earlier flushes are waited for only if they may still be in flight
INF_FN void inf_copy_far(InfState& s, uint32_t op, uint32_t dist, uint32_t len) {
    uint8_t* ring = s.sc->ring;
    W_VEC(uint32_t, v);
    if (op - dist + len > s.clean) { W_FENCE(); s.clean = s.flushed; }
    for (uint32_t done = 0; done < len; done += 64u) {
        const uint32_t n = len - done < 64u ? len - done : 64u;
        W_FOR { if ((uint32_t)W_LANE < n) V(v) = s.out[op + done + (uint32_t)W_LANE - dist]; }
        W_FOR { if ((uint32_t)W_LANE < n) ring[inf_ridx(s, op + done + (uint32_t)W_LANE)] = (uint8_t)V(v); }
    }
}
INF_FN int inf_match(InfState& s, uint32_t dist, uint32_t len) {
    if (dist > s.pos) return INF_E_DIST;
    if (s.pos + len > s.out_cap) return INF_E_OUTPUT;
    if (dist + len + INF_STEP_CAP <= INF_RING) inf_copy_near(s, s.pos, dist, len);
    else inf_copy_far(s, s.pos, dist, len);
    s.pos += len;
    return 0;
}

// ---- canonical Huffman ----------------------------------------------------------------------------------------------------------------------------
// slow path: a code longer than its fast table (counts per length in registers, symbols in canonical order in scratch)
INF_FN int inf_decode_slow(InfState& s, const uint16_t* counts, const uint16_t* symbol) {
    uint32_t bits = inf_peek(s, s.bp);
    int code = 0, first = 0, index = 0;
#pragma unroll 1
    for (int len = 1; len <= INF_MAXBITS; len++) {
        code |= (int)(bits & 1u); bits >>= 1;
#ifdef INF_HOST
        const int count = counts[len];
#else
        const int count = __builtin_amdgcn_readfirstlane((int)counts[len]);
#endif
        if (code - count < first) {
            s.bp += (uint32_t)len;
#ifdef INF_HOST
            return (int)symbol[index + (code - first)];
#else
            return __builtin_amdgcn_readfirstlane((int)symbol[index + (code - first)]);
#endif
        }
        index += count; first += count;
        first <<= 1; code <<= 1;
    }
    return INF_E_SYMBOL;
}

// Code lengths len[0..n) (scratch) -> counts per length, symbols in canonical order, fast table of `fb` bits; simple_below: symbols below it carry
// INF_SIMPLE.  All lanes work: lane l owns symbols l, l + 64, ...; the rank of a symbol among those of its length comes from ballots.
// Returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed; *used = symbols with a code.
INF_FN int inf_build(InfState& s, const uint8_t* len, int n, uint16_t* counts, uint16_t* symbol, uint16_t* fast, int fb, int simple_below, int* used) {
    W_VEC2(uint32_t, L, 5);
#pragma unroll
    for (int r = 0; r < 5; r++) { W_FOR { const int i = r * 64 + W_LANE; V2(L, r) = i < n ? (uint32_t)len[i] : 0u; } }
    W_FOR { for (int k = W_LANE; k < (1 << (fb - 1)); k += 64) reinterpret_cast<uint32_t*>(fast)[k] = 0u; }
    int left = 1;
    uint32_t code = 0, base = 0;
#pragma unroll 1
    for (int l = 1; l <= INF_MAXBITS; l++) {
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            if (r * 64 >= n) break;
            uint64_t m;
            W_BALLOT(m, V2(L, r) == (uint32_t)l);
            if (m) {
                W_FOR {
                    if (V2(L, r) == (uint32_t)l) {
                        const uint32_t rank = cnt + W_RANK(m), sym = (uint32_t)(r * 64 + W_LANE);
                        symbol[base + rank] = (uint16_t)sym;
                        if (l <= fb) {
                            const uint32_t rev = inf_bitrev(code + rank) >> (32 - l);
                            const uint16_t e = (uint16_t)(((uint32_t)l << 10) | ((int)sym < simple_below ? INF_SIMPLE : 0u) | sym);
                            for (uint32_t k = rev; k < (1u << fb); k += 1u << l) fast[k] = e;
                        }
                    }
                }
                cnt += (uint32_t)__builtin_popcountll(m);
            }
        }
        W_FOR { if (W_LANE == 0) counts[l] = (uint16_t)cnt; }
        left <<= 1; left -= (int)cnt;
        if (left < 0) return left;
        base += cnt;
        code = (code + cnt) << 1;
    }
    *used = (int)base;
    return left;
}

// one 64-offset lookup step: entry of the code that would start at bit bp + lane, for a table of fb bits
#define INF_WINDOW(s, fast, fb, E)                                                                                                         \
    {                                                                                                                                      \
        const uint32_t d0_ = (s).bp >> 5, sh_ = (s).bp & 31u;                                                                              \
        uint32_t w0_, w1_, w2_, w3_;                                                                                                       \
        INF_WORDS4(s, d0_, w0_, w1_, w2_, w3_)                                                                                             \
        W_FOR {                                                                                                                            \
            const uint32_t b_ = sh_ + (uint32_t)W_LANE, i_ = b_ >> 5, f_ = b_ & 31u;                                                       \
            const uint32_t lo_ = i_ == 0u ? w0_ : (i_ == 1u ? w1_ : w2_), hi_ = i_ == 0u ? w1_ : (i_ == 1u ? w2_ : w3_);                   \
            const uint32_t bits_ = (uint32_t)((((uint64_t)hi_ << 32) | lo_) >> f_);                                                        \
            V(E) = (fast)[bits_ & ((1u << (fb)) - 1u)];                                                                                    \
        }                                                                                                                                  \
    }

// length symbol (257 + i) -> base length and extra bits; distance symbol -> base distance and extra bits (RFC 1951 3.2.5 in closed form)
INF_FN uint32_t inf_len_extra(uint32_t i) { return i < 8u || i >= 28u ? 0u : (i - 4u) >> 2; }
INF_FN uint32_t inf_len_base(uint32_t i, uint32_t xb) { return i < 8u ? 3u + i : (i >= 28u ? 258u : 3u + ((4u + (i & 3u)) << xb)); }
INF_FN uint32_t inf_dist_extra(uint32_t ds) { return ds < 4u ? 0u : (ds - 2u) >> 1; }
INF_FN uint32_t inf_dist_base(uint32_t ds, uint32_t xb) { return ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << xb); }

// base values and extra-bit counts of the length / distance symbols (RFC 1951 3.2.5) as two LDS tables: they share their bytes with the code lengths of the
// header, so they are written again after the tables of every block are built
INF_FN void inf_token_luts(InfState& s) {
    InfScratch& sc = *s.sc;
    W_FOR {
        if (W_LANE < 32) {
            const uint32_t i = (uint32_t)W_LANE, lx = inf_len_extra(i), dx = inf_dist_extra(i);
            sc.len_lut[i] = (uint16_t)(i <= 28u ? inf_len_base(i, lx) | (lx << 9) : 0u);
            sc.dist_lut[i] = i < 30u ? inf_dist_base(i, dx) | (dx << 16) : 0u;
        }
    }
}

// emission of one step's tokens (mask MT over the lanes; T / DV / E per token lane; EX = output offset of each token inside the step, acc = bytes of the step)
// one output byte per lane when the step produces at most 64 bytes: which token a byte belongs to comes from a scatter + running maximum, a match byte whose
// source lies before the step is gathered from the ring, one whose source is a byte of this very step follows it by pointer doubling
template <class VecT>
INF_FN int inf_emit_bytes(InfState& s, uint64_t MT, const VecT& T, const VecT& DV, const VecT& E, const VecT& EX, uint32_t acc) {
    InfScratch& sc = *s.sc;
    W_VEC(uint32_t, tk); W_VEC(uint32_t, PK); W_VEC(uint32_t, pk); W_VEC(uint32_t, val); W_VEC(uint32_t, ref); W_VEC(uint32_t, un); W_VEC(uint32_t, idx);
    W_FOR { if (W_LANE < 16) reinterpret_cast<uint32_t*>(sc.tokmap)[W_LANE] = 0u; }
    W_FOR { if ((MT >> W_LANE) & 1ull) sc.tokmap[V(EX)] = (uint8_t)(W_LANE + 1); }
    W_FOR { V(tk) = (uint32_t)W_LANE < acc ? (uint32_t)sc.tokmap[W_LANE] : 0u; }
    W_INCL_MAX_SCAN(tk);
    W_FOR {
        V(PK) = (V(DV) & 0xffffu) | (V(EX) << 16) | ((V(T) & INF_T_MATCH) ? 1u << 22 : 0u) | ((V(T) & INF_T_FAR) ? 1u << 23 : 0u) | ((V(E) & 255u) << 24);
        V(idx) = (V(tk) - 1u) & 63u;
    }
    W_BPERMUTE(pk, PK, idx);
    uint64_t bad, wait;
    W_BALLOT(bad, (uint32_t)W_LANE < acc && ((V(pk) >> 22) & 1u) && (V(pk) & 0xffffu) > s.pos + ((V(pk) >> 16) & 63u));
    if (bad) return INF_E_DIST;                                   // a distance reaching in front of the output
    // far sources come back from the flushed output (their distance puts them in front of everything pending): wait for those stores if they may be in flight
    W_BALLOT(wait, (uint32_t)W_LANE < acc && ((V(pk) >> 23) & 1u) && s.pos + (uint32_t)W_LANE - (V(pk) & 0xffffu) >= s.clean);
    if (wait) { W_FENCE(); s.clean = s.flushed; }
    W_FOR {
        const uint32_t dist = V(pk) & 0xffffu;
        const bool m = (uint32_t)W_LANE < acc && ((V(pk) >> 22) & 1u);
        V(val) = V(pk) >> 24;
        V(un) = 0u; V(ref) = 0u;
        if (m) {
            if (dist > (uint32_t)W_LANE) {                                                                    // source in front of the step
                if ((V(pk) >> 23) & 1u) V(val) = s.out[s.pos + (uint32_t)W_LANE - dist];
                else V(val) = sc.ring[inf_ridx(s, s.pos + (uint32_t)W_LANE - dist)];
            } else { V(un) = 1u; V(ref) = (uint32_t)W_LANE - dist; }                                          // source is a byte of this step
        }
    }
    for (;;) {
        uint64_t U;
        W_BALLOT(U, V(un) != 0u);
        if (!U) break;
        W_VEC(uint32_t, vv); W_VEC(uint32_t, v2); W_VEC(uint32_t, r2);
        W_FOR { V(vv) = V(val) | (V(un) << 8); }
        W_BPERMUTE(v2, vv, ref);
        W_BPERMUTE(r2, ref, ref);
        W_FOR { if (V(un)) { if (!((V(v2) >> 8) & 1u)) { V(val) = V(v2) & 255u; V(un) = 0u; } else V(ref) = V(r2); } }
    }
    W_FOR { if ((uint32_t)W_LANE < acc) sc.ring[inf_ridx(s, s.pos + (uint32_t)W_LANE)] = (uint8_t)V(val); }
    return 0;
}

// the same for a step that decoded SEVERAL windows (window w = bit offsets 64 w .. 64 w + 63 of the step; mask / T / DV / E / EX per window, EX already counted from
// the start of the step): at most 64 output bytes in all, one per lane - and therefore at most 64 tokens: they are ranked in output order (window by window,
// lane by lane), their parameters go to a list in LDS, and every output byte finds its token through the rank
#ifndef INF_MAX_WIN
#define INF_MAX_WIN 4
#endif
template <class VecT>
INF_FN int inf_emit_bytes_multi(InfState& s, int nw, const uint64_t* MW, VecT* TW, VecT* DW, VecT* EW, VecT* XW, uint32_t acc) {
    InfScratch& sc = *s.sc;
    W_VEC(uint32_t, tk); W_VEC(uint32_t, pk); W_VEC(uint32_t, val); W_VEC(uint32_t, ref); W_VEC(uint32_t, un);
    W_FOR { if (W_LANE < 16) reinterpret_cast<uint32_t*>(sc.tokmap)[W_LANE] = 0u; }
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < INF_MAX_WIN; w++) {
        if (w < nw) {
            const uint64_t M = MW[w];
            W_FOR {
                if ((M >> W_LANE) & 1ull) {
                    const uint32_t r = base + W_RANK(M);
                    sc.tokmap[V2(XW, w)] = (uint8_t)(r + 1u);
                    sc.pk_list[r & 63u] = (V2(DW, w) & 0xffffu) | (V2(XW, w) << 16) | ((V2(TW, w) & INF_T_MATCH) ? 1u << 22 : 0u) | ((V2(TW, w) & INF_T_FAR) ? 1u << 23 : 0u) |
                                          ((V2(EW, w) & 255u) << 24);
                }
            }
            base += (uint32_t)__builtin_popcountll(M);
        }
    }
    W_FOR { V(tk) = (uint32_t)W_LANE < acc ? (uint32_t)sc.tokmap[W_LANE] : 0u; }
    W_INCL_MAX_SCAN(tk);
    W_FOR { V(pk) = sc.pk_list[(V(tk) - 1u) & 63u]; }
    uint64_t bad, wait;
    W_BALLOT(bad, (uint32_t)W_LANE < acc && ((V(pk) >> 22) & 1u) && (V(pk) & 0xffffu) > s.pos + ((V(pk) >> 16) & 63u));
    if (bad) return INF_E_DIST;                                   // a distance reaching in front of the output
    W_BALLOT(wait, (uint32_t)W_LANE < acc && ((V(pk) >> 23) & 1u) && s.pos + (uint32_t)W_LANE - (V(pk) & 0xffffu) >= s.clean);
    if (wait) { W_FENCE(); s.clean = s.flushed; }
    W_FOR {
        const uint32_t dist = V(pk) & 0xffffu;
        const bool m = (uint32_t)W_LANE < acc && ((V(pk) >> 22) & 1u);
        V(val) = V(pk) >> 24;
        V(un) = 0u; V(ref) = 0u;
        if (m) {
            if (dist > (uint32_t)W_LANE) {                                                                    // source in front of the step
                if ((V(pk) >> 23) & 1u) V(val) = s.out[s.pos + (uint32_t)W_LANE - dist];
                else V(val) = sc.ring[inf_ridx(s, s.pos + (uint32_t)W_LANE - dist)];
            } else { V(un) = 1u; V(ref) = (uint32_t)W_LANE - dist; }                                          // source is a byte of this step
        }
    }
    for (;;) {
        uint64_t U;
        W_BALLOT(U, V(un) != 0u);
        if (!U) break;
        W_VEC(uint32_t, vv); W_VEC(uint32_t, v2); W_VEC(uint32_t, r2);
        W_FOR { V(vv) = V(val) | (V(un) << 8); }
        W_BPERMUTE(v2, vv, ref);
        W_BPERMUTE(r2, ref, ref);
        W_FOR { if (V(un)) { if (!((V(v2) >> 8) & 1u)) { V(val) = V(v2) & 255u; V(un) = 0u; } else V(ref) = V(r2); } }
    }
    W_FOR { if ((uint32_t)W_LANE < acc) sc.ring[inf_ridx(s, s.pos + (uint32_t)W_LANE)] = (uint8_t)V(val); }
    return 0;
}

// every lane: the WHOLE token that would start at its bit offset, from the 32 input bits x there and the literal/length entry ee of its first code (literal, or
// length + extra bits + distance code + extra bits: base values and extra-bit counts come from two small LDS tables).  32-bit arithmetic: a token of more than
// 32 bits (long distance codes with many extra bits) is left to the serial path (INF_T_STOP)
#define INF_TOKEN(sc, x_, ee_, t_out, dv_out)                                                                               \
    {                                                                                                                       \
        const uint32_t x = (x_), ee = (ee_), cl = ee >> 10;                                                                 \
        uint32_t t, dv = 0;                                                                                                 \
        if (ee & INF_SIMPLE) t = cl | (1u << 8);                                                                            \
        else {                                                                                                              \
            const uint32_t i = (ee & 511u) - 257u;                                   /* 0..28 for a length symbol */         \
            const uint32_t ll = (sc).len_lut[i & 31u];                                                                      \
            const uint32_t xb = ll >> 9;                                                                                    \
            const uint32_t len = (ll & 511u) + ((x >> cl) & ((1u << xb) - 1u));                                             \
            const uint32_t p2 = cl + xb;                                             /* <= 15 */                            \
            const uint32_t de = (sc).fast_d[(x >> p2) & ((1u << INF_FAST_D) - 1u)];                                         \
            const uint32_t dl = (sc).dist_lut[de & 31u];                                                                    \
            const uint32_t dxb = dl >> 16, p3 = p2 + (de >> 10);                       /* p3 <= 23 */                        \
            dv = (dl & 0xffffu) + ((x >> p3) & ((1u << dxb) - 1u));                                                         \
            const bool ok = ee != 0u && i <= 28u && de != 0u && (de & 511u) < 30u && p3 + dxb <= 32u;                       \
            t = ok ? ((p3 + dxb) | (len << 8) | INF_T_MATCH | (dv + len + INF_STEP_CAP <= INF_RING ? 0u : INF_T_FAR)) : INF_T_STOP; \
        }                                                                                                                   \
        t_out = t; dv_out = dv;                                                                                             \
    }

INF_FN int inf_codes(InfState& s) {
    InfScratch& sc = *s.sc;
    bool match_mode = false;                                       // the step before produced matches: decode whole tokens right away
    bool wide = false;                                             // ... and few bytes per input bit (short matches between literals: base qualities): the
                                                                   // next step looks at TWO windows, 128 bits (see below)
    for (;;) {
        if (s.bp > 32u * s.in_words + 64u) return INF_E_INPUT;
        inf_sync_input(s);
        inf_maybe_flush(s);
        INF_PROF(s, 8)
        // every lane: the 32 input bits from its offset on, and the literal/length entry of the code that would start there
        W_VEC(uint32_t, XL); W_VEC(uint32_t, E);
        {
            const uint32_t d0 = s.bp >> 5, sh = s.bp & 31u;
            uint32_t w0, w1, w2, w3;
            INF_WORDS4(s, d0, w0, w1, w2, w3)
            W_FOR {
                const uint32_t b = sh + (uint32_t)W_LANE, i = b >> 5, f = b & 31u;
                const uint32_t a0 = i == 0u ? w0 : (i == 1u ? w1 : w2), a1 = i == 0u ? w1 : (i == 1u ? w2 : w3);
                V(XL) = (uint32_t)((((uint64_t)a1 << 32) | a0) >> f);
                V(E) = sc.fast_l[V(XL) & ((1u << INF_FAST_L) - 1u)];
            }
        }
        INF_PROF(s, 0) INF_COUNT(s, 9, 1)
        INF_STEP_STAT()
        uint32_t o = 0, e;
        if (!match_mode) {
            // phase A: follow the code starts through the 64 entries while they are literals
            uint64_t M = 0;
#define INF_HOP_A                                                                                                           \
            e = W_READLANE(E, o);                                                                                           \
            if (!(e & INF_SIMPLE)) break;                                                                                   \
            M |= 1ull << o;                                                                                                 \
            o += e >> 10;                                                                                                   \
            if (o >= 64u) { e = INF_SIMPLE; break; }              /* window used up; the next step continues at bp + o */
            for (;;) { INF_HOP_A INF_HOP_A INF_HOP_A INF_HOP_A }        // (four hops per taken branch)
#undef INF_HOP_A
            const uint32_t nlit = (uint32_t)__builtin_popcountll(M);
            if (nlit) {
                if (s.pos + nlit > s.out_cap) return INF_E_OUTPUT;
                W_FOR { if ((M >> W_LANE) & 1ull) sc.ring[inf_ridx(s, s.pos + W_RANK(M))] = (uint8_t)V(E); }
                s.pos += nlit;
            }
            INF_PROF(s, 1) INF_COUNT(s, 13, nlit)
            if (e & INF_SIMPLE) { s.bp += o; continue; }
        } else e = W_READLANE(E, 0u);
        if (match_mode || (e != 0u && (e & 511u) >= 257u && (e & 511u) <= 285u)) {
            // phase B: every lane decodes the WHOLE token that would start at its offset (literal, or length + extra bits + distance code + extra bits: base
            // values and extra-bit counts come from two small LDS tables), then the chain runs on over literals and matches alike
            W_VEC(uint32_t, T); W_VEC(uint32_t, DV);
            W_FOR { INF_TOKEN(sc, V(XL), V(E), V(T), V(DV)) }
            INF_PROF(s, 2) INF_COUNT(s, 10, 1)
#ifndef INF_NO_WIDE
            if (wide && o == 0u) {
                // Several windows per step.  A step costs about the same whatever it emits: the flush test, the scans and LDS round trips of the emission.
                // Literal-heavy streams (base qualities: short matches between literals, ~12 bytes per 64 bits) use a fifth of the emission's 64 lanes, so the
                // step takes the next 64 bit offsets as well - lane l decodes the tokens at offsets l, 64 + l, 128 + l ... -, the chain runs on through
                // the windows, and ONE emission writes up to 64 bytes.  Windows are added while their output fits (and is expected to fit) into the 64
                // bytes; one that does not is dropped and the next step starts there.  With a single window left the step goes on as an ordinary one.
                W_VEC2(uint32_t, TW, INF_MAX_WIN); W_VEC2(uint32_t, DW, INF_MAX_WIN); W_VEC2(uint32_t, EW, INF_MAX_WIN); W_VEC2(uint32_t, XW, INF_MAX_WIN);
                uint64_t MW[INF_MAX_WIN];
                W_FOR { V2(TW, 0) = V(T); V2(DW, 0) = V(DV); V2(EW, 0) = V(E); }
                uint32_t p = 0, total = 0, bits = 0;               // chain position inside the current window; bytes / input bits of the windows taken so far
                int nw = 0;
                uint64_t any_match = 0;
#pragma unroll
                for (int w = 0; w < INF_MAX_WIN; w++) {
                    if (w > 0) {
                        const uint32_t dw = (s.bp >> 5) + 2u * (uint32_t)w, sh = s.bp & 31u;
                        uint32_t w0, w1, w2, w3;
                        INF_WORDS4(s, dw, w0, w1, w2, w3)
                        W_FOR {
                            const uint32_t b = sh + (uint32_t)W_LANE, i = b >> 5, f = b & 31u;
                            const uint32_t a0 = i == 0u ? w0 : (i == 1u ? w1 : w2), a1 = i == 0u ? w1 : (i == 1u ? w2 : w3);
                            const uint32_t xl = (uint32_t)((((uint64_t)a1 << 32) | a0) >> f);
                            V2(EW, w) = sc.fast_l[xl & ((1u << INF_FAST_L) - 1u)];
                            INF_TOKEN(sc, xl, V2(EW, w), V2(TW, w), V2(DW, w))
                        }
                    }
                    uint64_t M = 0;
                    uint32_t tt = 0;
                    bool stopped = false;
                    for (;;) {
                        tt = W_READLANE(V2W(TW, w), p);
                        if (tt & INF_T_STOP) { stopped = true; break; }
                        M |= 1ull << p;
                        p += tt & 63u;
                        if (p >= 64u) break;
                    }
                    if (!M) break;                                  // the window starts with a token the lanes do not decode: the serial path's
                    W_VEC(uint32_t, OL);
                    uint32_t acc;
                    W_FOR { V(OL) = ((M >> W_LANE) & 1ull) ? (V2(TW, w) >> 8) & 511u : 0u; }
                    W_EXCL_SCAN(V2W(XW, w), OL, acc);
                    if (total + acc > 64u) break;                   // does not fit: dropped
                    W_FOR { V2(XW, w) += total; }
                    MW[w] = M;
                    total += acc;
                    bits = 64u * (uint32_t)w + p;                    // (a chain that stopped: p is where the next step starts)
                    nw = w + 1;
                    uint64_t MMw;
                    W_BALLOT(MMw, ((M >> W_LANE) & 1ull) && (V2(TW, w) & INF_T_MATCH));
                    any_match |= MMw;
                    if (stopped) break;
                    p -= 64u;
                    if (total + total / (uint32_t)(w + 1) > 64u) break;         // the next window is not expected to fit
                }
                if (nw >= 2) {
                    if (s.pos + total > s.out_cap) return INF_E_OUTPUT;
                    const int rc = inf_emit_bytes_multi(s, nw, MW, TW, DW, EW, XW, total);
                    if (rc) return rc;
                    s.pos += total;
                    s.bp += bits;
                    match_mode = any_match != 0ull;
                    wide = match_mode && 2u * total <= bits;              // still at most half a byte per input bit
#ifdef INF_HOST
                    for (int w = 0; w < nw; w++) { W_FOR { if ((MW[w] >> W_LANE) & 1ull) INF_STAT(0, (V2(TW, w) & INF_T_MATCH) ? (V2(TW, w) >> 8) & 511u : 0u, V2(DW, w)); } }
#endif
#if defined(INF_PROFILE) && !defined(INF_HOST)
                    { uint32_t ntok = 0; for (int w = 0; w < nw; w++) ntok += (uint32_t)__builtin_popcountll(MW[w]); INF_PROF(s, 4) INF_COUNT(s, 14, ntok) }
#endif
                    INF_WIDE_STAT(nw)
                    continue;
                }
                INF_WIDE_STAT(0)
                wide = false;                                      // (the ordinary step below decodes the first window again from T / DV, which are untouched)
            }
#endif
            uint64_t MT = 0;
            uint32_t o2 = o, t = 0;
#define INF_HOP_B                                                                                                           \
            t = W_READLANE(T, o2);                                                                                          \
            if (t & INF_T_STOP) break;                                                                                      \
            MT |= 1ull << o2;                                                                                               \
            o2 += t & 63u;                                                                                                  \
            if (o2 >= 64u) break;
            for (;;) { INF_HOP_B INF_HOP_B INF_HOP_B INF_HOP_B }
#undef INF_HOP_B
            INF_PROF(s, 3)
            if (MT) {
                W_VEC(uint32_t, OL); W_VEC(uint32_t, EX);
                W_FOR { V(OL) = ((MT >> W_LANE) & 1ull) ? (V(T) >> 8) & 511u : 0u; }
                uint32_t acc;
                W_EXCL_SCAN(EX, OL, acc);
                if (acc > INF_STEP_CAP) {
                    // too much output for one step (a window of maximal matches): keep the tokens that fit - a prefix of the chain
                    uint64_t keep;
                    W_BALLOT(keep, ((MT >> W_LANE) & 1ull) && V(EX) + V(OL) <= INF_STEP_CAP);
                    const uint64_t dropped = MT & ~keep;
                    o2 = (uint32_t)__builtin_ctzll(dropped);
                    MT = keep;
                    W_FOR { if (!((MT >> W_LANE) & 1ull)) V(OL) = 0u; }
                    W_EXCL_SCAN(EX, OL, acc);
                }
                if (s.pos + acc > s.out_cap) return INF_E_OUTPUT;
                uint64_t MM;
                W_BALLOT(MM, ((MT >> W_LANE) & 1ull) && (V(T) & INF_T_MATCH));
                if (acc <= 64u) {
                    const int rc = inf_emit_bytes(s, MT, T, DV, E, EX, acc);
                    if (rc) return rc;
                    INF_PROF(s, 4)
                } else {
                    W_FOR { if (((MT >> W_LANE) & 1ull) && !(V(T) & INF_T_MATCH)) sc.ring[inf_ridx(s, s.pos + V(EX))] = (uint8_t)V(E); }
                    uint64_t mm = MM;
                    while (mm) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(mm);
                        mm &= mm - 1ull;
                        const uint32_t len = (W_READLANE(T, k) >> 8) & 511u, dist = W_READLANE(DV, k), op = s.pos + W_READLANE(EX, k);
                        if (dist > op) return INF_E_DIST;
                        if (W_READLANE(T, k) & INF_T_FAR) inf_copy_far(s, op, dist, len); else inf_copy_near(s, op, dist, len);
                    }
                    INF_PROF(s, 5) INF_COUNT(s, 12, 1)
                }
                s.pos += acc;
                s.bp += o2;
                match_mode = MM != 0ull;
                wide = match_mode && o2 >= 48u && 2u * acc <= o2;       // few bytes per input bit, and the window was used up: two windows next time
                W_FOR { if ((MT >> W_LANE) & 1ull) INF_STAT(0, (V(T) & INF_T_MATCH) ? (V(T) >> 8) & 511u : 0u, V(DV)); }
                INF_COUNT(s, 14, __builtin_popcountll(MT))
                continue;                                               // (a token the chain stopped at starts the next step)
            }
            // the very first token is not a plain near match: the serial path below decodes it
            e = W_READLANE(E, o);
        }
        match_mode = false; wide = false;
        s.bp += o;
        int sym;
        if (e) { sym = (int)(e & 511u); s.bp += e >> 10; }
        else {
            sym = inf_decode_slow(s, sc.cnt_l, sc.lsym);
            if (sym < 0) return sym;
        }
        INF_COUNT(s, 11, 1)
        if (sym < 256) {
            if (s.pos >= s.out_cap) return INF_E_OUTPUT;
            W_FOR { if (W_LANE == 0) sc.ring[inf_ridx(s, s.pos)] = (uint8_t)sym; }
            s.pos++;
            INF_PROF(s, 6)
            continue;
        }
        if (sym == 256) return 0;
        sym -= 257;
        if (sym >= 29) return INF_E_SYMBOL;
        const uint32_t lxb = inf_len_extra((uint32_t)sym);
        const uint32_t len = inf_len_base((uint32_t)sym, lxb) + inf_bits(s, (int)lxb);
        int ds;
        {
            const uint32_t idx = inf_peek(s, s.bp) & ((1u << INF_FAST_D) - 1u);
#ifdef INF_HOST
            const uint32_t de = sc.fast_d[idx];
#else
            const uint32_t de = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.fast_d[idx]);
#endif
            if (de) { ds = (int)(de & 511u); s.bp += de >> 10; }
            else { ds = inf_decode_slow(s, sc.cnt_d, sc.dsym); if (ds < 0) return ds; }
        }
        if (ds >= 30) return INF_E_SYMBOL;
        const uint32_t dxb = inf_dist_extra((uint32_t)ds);
        const uint32_t dist = inf_dist_base((uint32_t)ds, dxb) + inf_bits(s, (int)dxb);
        INF_STAT(1, len, dist);
        const int rc = inf_match(s, dist, len);
        if (rc) return rc;
        INF_PROF(s, 6)
    }
}

// the code lengths of a dynamic block: nlen + ndist values coded with the code-length code (RFC 1951 3.2.7) -> sc.len[0 .. nlen + ndist)
INF_FN int inf_code_lengths(InfState& s, int total) {
    InfScratch& sc = *s.sc;
    int index = 0;
    while (index < total) {
        if (s.bp > 32u * s.in_words + 64u) return INF_E_INPUT;
        inf_sync_input(s);
        W_VEC(uint32_t, E);
        INF_WINDOW(s, sc.fast_c, INF_FAST_C, E);
        uint64_t M = 0;
        uint32_t o = 0, e;
        int room = total - index;
        for (;;) {
            e = W_READLANE(E, o);
            if (!(e & INF_SIMPLE) || room == 0) break;
            M |= 1ull << o; room--;
            o += e >> 10;
            if (o >= 64u) { e = INF_SIMPLE; break; }
        }
        const int k = __builtin_popcountll(M);
        if (k) { W_FOR { if ((M >> W_LANE) & 1ull) sc.len[index + (int)W_RANK(M)] = (uint8_t)(V(E) & 31u); } index += k; }
        s.bp += o;
        if ((e & INF_SIMPLE) || index >= total) continue;
        int sym;
        if (e) { sym = (int)(e & 511u); s.bp += e >> 10; }
        else { sym = inf_decode_slow(s, sc.cnt_c, sc.csym); if (sym < 0) return sym; }
        if (sym < 16) { W_FOR { if (W_LANE == 0) sc.len[index] = (uint8_t)sym; } index++; continue; }
        uint32_t prev = 0, rep;
        if (sym == 16) {
            if (index == 0) return INF_E_CODELEN;
#ifdef INF_HOST
            prev = sc.len[index - 1];
#else
            prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.len[index - 1]);
#endif
            rep = 3u + inf_bits(s, 2);
        } else if (sym == 17) rep = 3u + inf_bits(s, 3);
        else rep = 11u + inf_bits(s, 7);
        if (index + (int)rep > total) return INF_E_CODELEN;
        W_FOR { for (uint32_t r = (uint32_t)W_LANE; r < rep; r += 64u) sc.len[index + (int)r] = (uint8_t)prev; }
        index += (int)rep;
    }
    return 0;
}

// one raw DEFLATE stream: `payload` (any alignment, in_bytes long), out_cap = ISIZE.  Returns the number of bytes produced (== ISIZE for a
// sound block) or a negative INF_E_*.  Device: every lane of the wave must call, with wave-uniform arguments; `sc` is that wave's own scratch.
INF_FN int inflate_raw(const uint8_t* payload, uint32_t in_bytes, uint8_t* out, uint32_t out_cap, InfScratch& sc) {
    InfState s;
    // the bit reader works on aligned 32-bit words: the up to 3 bytes in front of the payload are skipped
    const uint32_t skip = (uint32_t)(reinterpret_cast<uintptr_t>(payload) & 3u);
    in_bytes += skip;
    s.in = reinterpret_cast<const uint32_t*>(payload - skip); s.in_words = (in_bytes + 3u) / 4u; s.bp = 8u * skip;
    s.out = out; s.out_cap = out_cap; s.pos = 0; s.flushed = 0; s.clean = 0; s.mis = (uint32_t)(reinterpret_cast<uintptr_t>(out) & 15u);
    s.sc = &sc;
#if defined(INF_PROFILE) && !defined(INF_HOST)
    for (int i = 0; i < 16; i++) s.prof[i] = 0;
    s.t_last = __builtin_readcyclecounter();
#endif
    inf_load_chunks(s, 0);
    int used = 0;
    for (;;) {
        if (s.bp > 32u * s.in_words + 64u) return INF_E_INPUT;
        inf_sync_input(s);
        const uint32_t last = inf_bits(s, 1), type = inf_bits(s, 2);
        if (type == 0) {
            // stored: skip to the byte boundary, LEN, ~LEN, LEN bytes
            s.bp = (s.bp + 7u) & ~7u;
            const uint32_t len = inf_bits(s, 16), nlen = inf_bits(s, 16);
            if (len != (~nlen & 0xffffu)) return INF_E_STORED;
            if (s.pos + len > s.out_cap) return INF_E_OUTPUT;
            const uint32_t at = s.bp >> 3;
            if (at + len > in_bytes) return INF_E_INPUT;
            // 1 KiB per trip: every lane assembles 4 x 4 bytes from aligned input words, all loads issued before the first is used (the input may be host
            // memory behind PCIe: one round trip per trip)
            const uint32_t sh8 = (at & 3u) * 8u;
            const uint32_t* wsrc = s.in + (at >> 2);
            for (uint32_t done = 0; done < len; done += 1024u) {
                const uint32_t n = len - done < 1024u ? len - done : 1024u;
                W_VEC2(uint32_t, v, 4);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    W_FOR {
                        const uint32_t i4 = 256u * (uint32_t)r + 4u * (uint32_t)W_LANE;            // first of this lane's 4 bytes in the trip
                        if (i4 < n) {
                            const uint32_t wi = (done + i4) >> 2;
                            const uint32_t lo = wsrc[wi], hi = sh8 ? wsrc[wi + 1u] : 0u;                 // (in_words covers at + len: the word behind the last byte may be padding)
                            V2(v, r) = sh8 ? (lo >> sh8) | (hi << (32u - sh8)) : lo;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    W_FOR {
                        const uint32_t i4 = 256u * (uint32_t)r + 4u * (uint32_t)W_LANE;
                        for (uint32_t k = 0; k < 4u; k++) if (i4 + k < n) sc.ring[inf_ridx(s, s.pos + i4 + k)] = (uint8_t)(V2(v, r) >> (8u * k));
                    }
                }
                s.pos += n;
                inf_flush(s, s.pos - ((s.pos + s.mis) & 15u));
            }
            s.bp += 8u * len;
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                W_FOR {
                    for (int i = W_LANE; i < 288; i += 64) sc.len[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
                    if (W_LANE < 30) sc.len[288 + W_LANE] = 5;
                }
                (void)inf_build(s, sc.len, 288, sc.cnt_l, sc.lsym, sc.fast_l, INF_FAST_L, 256, &used);
                (void)inf_build(s, sc.len + 288, 30, sc.cnt_d, sc.dsym, sc.fast_d, INF_FAST_D, 0, &used);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)inf_bits(s, 5) + 257, ndist = (int)inf_bits(s, 5) + 1, ncode = (int)inf_bits(s, 4) + 4;
                if (nlen > 286 || ndist > 30) return INF_E_CODELEN;
                // the code-length code: 19 symbols of up to 7 bits, 3 bits each in the header (57 bits: two peeks)
                W_FOR { if (W_LANE < 19) sc.len[W_LANE] = 0; }
                {
                    const uint32_t a = inf_peek(s, s.bp), b = inf_peek(s, s.bp + 30u);
                    W_FOR {
                        if (W_LANE < ncode) {
                            const uint32_t v3 = W_LANE < 10 ? (a >> (3 * W_LANE)) & 7u : (b >> (3 * (W_LANE - 10))) & 7u;
                            sc.len[order[W_LANE]] = (uint8_t)v3;
                        }
                    }
                    s.bp += 3u * (uint32_t)ncode;
                }
                int err = inf_build(s, sc.len, 19, sc.cnt_c, sc.csym, sc.fast_c, INF_FAST_C, 16, &used);
                if (err != 0) return INF_E_CODELEN;                      // complete code required here
                err = inf_code_lengths(s, nlen + ndist);
                if (err) return err;
#ifdef INF_HOST
                const int eob_len = sc.len[256];
#else
                const int eob_len = __builtin_amdgcn_readfirstlane((int)sc.len[256]);
#endif
                if (eob_len == 0) return INF_E_CODELEN;                  // no end-of-block code
                err = inf_build(s, sc.len + nlen, ndist, sc.cnt_d, sc.dsym, sc.fast_d, INF_FAST_D, 0, &used);
                if (err < 0 || (err > 0 && used != 1)) return INF_E_OVERSUB;   // incomplete only allowed for a single distance code
                err = inf_build(s, sc.len, nlen, sc.cnt_l, sc.lsym, sc.fast_l, INF_FAST_L, 256, &used);
                if (err < 0 || (err > 0 && used != 1)) return INF_E_OVERSUB;
            }
            inf_token_luts(s);
            INF_PROF(s, 7)
            const int rc = inf_codes(s);
            if (rc) return rc;
        } else return INF_E_BLOCKTYPE;
        if (last) break;
    }
    if ((s.bp + 7u) / 8u > in_bytes) return INF_E_INPUT;
    inf_flush(s, s.pos);
#if defined(INF_PROFILE) && !defined(INF_HOST)
    INF_PROF(s, 8)
    s.prof[15] = s.pos;
    if (W_LANE == 0) for (int i = 0; i < 16; i++) atomicAdd(&g_inf_prof[i], (unsigned long long)s.prof[i]);
#endif
    return (int)s.pos;
}
