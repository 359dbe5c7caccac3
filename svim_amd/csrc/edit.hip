// edit.hip - global (Needleman-Wunsch, unit cost) edit distance between insertion haplotypes on the GPU.
//
// Replaces compute_haplotype_edit_distance (src/svim/SVIM_clustering.py:32-45): the reference builds
//   hap_k = ref[ws:start_k] + inserted_k + ref[start_k:we]   (window = min/max start -+ 100, clipped to the contig)
// as Python strings with four FASTA fetches per pair and calls edlib.align(h1, h2)["editDistance"].
//
// Pipeline (all exact; what varies with the data is only the ROUTE a pair takes, never its result):
//   1. k_hap_words / scan / k_hap_pack   one 4-bit packed record per insertion signature:
//                               ref[start-R, start) + inserted bases + ref[start, start+R), R = largest |start_a - start_b| of the
//                               work list + 100.  The haplotype of any pair a signature takes part in is a substring of its record
//                               (a nibble offset + a length), so nothing is materialised per pair.
//   2. k_edit_prep (wave/pair)  common prefix / suffix stripped on the packed words (8 symbols per lane and step), Hamming-style
//                               upper bound, class choice.  A pair is described by 32 bytes (PairDesc).
//   3. rounds.  Per round TWO fused launches: k_edit_bands (every band class, may fail) on a low-priority stream and
//      k_edit_fulls (every full-matrix class, never fails) on a high-priority one; a block looks up its class segment.
//        d_edit_band<Q>  (lane/pair, class 0: 32 diagonals; the pilot)   Ukkonen band in band-relative coordinates, the window
//                               slides one row per column
//        d_edit_stair    (lane/pair, classes 1..8: 96..512 diagonals)    window stands still for 32 columns, then drops a word - or two, and
//                               takes none in: it narrows by Ukkonen's cut-off against the largest distance the attempt certifies
//        d_edit_lane<Q>  (lane/pair)  whole pattern (<= 512 rows) in one lane, full matrix
//        d_edit_wide<G,Q>(G = 2..16 lanes/pair, Q = 10 / 12 / 14 / 16 words = 320..512 rows each)   full matrix up to 8192 rows, DPP hand-off between lanes
//        d_edit_full     (wave/pair)  64-lane systolic full matrix; cores beyond 16384 rows keep block state in a scratch area
//      All share one column update (MYERS_COLUMN: Myers/Hyyro bit-vector recurrence, add-with-carry chain, explicit v_bitop3),
//      in a 2-bit-plane (A/C/G/T only) and a 4-bit-plane (any BAM code) instantiation.
//      A band result d is exact iff floor((d-(n-m))/2) <= band margin; otherwise d is only an upper bound and the pair is appended
//      to the retry list of a wider class (2x..4x), consumed by the next round without re-sorting.  A round waits only for its
//      band launch, so retry rounds overlap the full-matrix work of earlier rounds.
//   4. k_edit_hist              divergence histogram of this call's pairs; calibrates the NEXT call's band speculation.
// No LDS, no MFMA: integer-ALU bound (reported as cell updates/s).  Instruction-rate facts the kernels are shaped by are
// measured by tools/micro/valu_ops.hip, valu_dep.hip and column_rate.hip (v_alignbit, v_addc_co and every instruction reading three
// different VGPRs are half rate; the column update costs 38 issue cycles per word-column and runs at 42-44 stand-alone).
#include "common.hpp"
#include "hostcopy.hpp"
#include <type_traits>

struct EditWork { uint32_t a, b; long long slot; };

struct Hap {
    const uint8_t* p0; const uint8_t* p1; const uint8_t* p2;
    int n0, n1, n2, len;
    __device__ __forceinline__ uint32_t at(int i) const {
        if (i < n0) return p0[i];
        i -= n0;
        if (i < n1) return p1[i];
        return p2[i - n1];
    }
    // symbols i..i+7 as one 4-bit packed word (symbols at or beyond `lim` read as 0): one unaligned 8-byte load when the run
    // lies inside a single piece, byte loads across a piece border
    __device__ __forceinline__ uint32_t pack8(int i, int lim) const {
        const int cnt = lim - i >= 8 ? 8 : (lim - i);
        if (cnt <= 0) return 0u;
        const uint8_t* q = nullptr;
        if (i + cnt <= n0) q = p0 + i;
        else if (i >= n0 && i + cnt <= n0 + n1) q = p1 + (i - n0);
        else if (i >= n0 + n1) q = p2 + (i - n0 - n1);
        if (q && cnt == 8) {
            unsigned long long x;
            __builtin_memcpy(&x, q, 8);
            x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
            x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
            x = (x | (x >> 16));
            return (uint32_t)x;
        }
        uint32_t w = 0;
        for (int k = 0; k < cnt; k++) w |= at(i + k) << (4 * k);
        return w;
    }
};

// A core inside the packed store: symbol i is nibble (i + sh) of w[] (sh = 0..7, the core starts anywhere inside its record).
// word(idx) = symbols 8 idx .. 8 idx + 7 as one word; every record is followed by two pad words, so w[idx + 1] is always readable.
struct Packed {
    const uint32_t* w; int sh4;
    __device__ __forceinline__ uint32_t word(int idx) const { return __builtin_amdgcn_alignbit(w[idx + 1], w[idx], sh4); }
    __device__ __forceinline__ uint32_t at(int i) const { const int k = i + (sh4 >> 2); return (w[k >> 3] >> ((k & 7) << 2)) & 15u; }
};
// 8 symbols starting at symbol s of a packed record (s >= -7: the store begins with one pad word)
__device__ __forceinline__ uint32_t fetch8(const uint32_t* w, int s) {
    const int i = s >> 3;
    return __builtin_amdgcn_alignbit(w[i + 1], w[i], (s & 7) << 2);
}

__device__ __forceinline__ void hap_fetch(const int64_t* g_off, const uint8_t* g_codes, int contig, long long a, long long b,
                                          const uint8_t*& p, int& n) {
    if (a < 0) a = 0;
    if (b < 0) b = 0;
    const long long len = g_off[contig + 1] - g_off[contig];
    if (b > len) b = len;
    if (a >= b) { p = g_codes; n = 0; return; }
    p = g_codes + g_off[contig] + a; n = (int)(b - a);
}

__device__ __forceinline__ Hap make_hap(const int64_t* g_off, const uint8_t* g_codes, int contig, long long start, const uint8_t* seq,
                                        int seq_len, long long ws, long long we) {
    Hap h;
    hap_fetch(g_off, g_codes, contig, ws, start, h.p0, h.n0);
    h.p1 = seq; h.n1 = seq_len;
    hap_fetch(g_off, g_codes, contig, start, we, h.p2, h.n2);
    h.len = h.n0 + h.n1 + h.n2;
    return h;
}

__device__ __forceinline__ Hap plain_hap(const uint8_t* s, int n) {
    Hap h; h.p0 = s; h.n0 = 0; h.p1 = s; h.n1 = n; h.p2 = s; h.n2 = 0; h.len = n; return h;
}

// One packed record per string: for a signature the stored string is  ref[start-R, start) + inserted bases + ref[start, start+R)
// (R = largest |start_a - start_b| over the work list + 100, clipped to the contig like FastaFile.fetch), so the haplotype of ANY
// pair the signature takes part in is a substring of its record; for the plain entry point a record is the string itself.
#define HAP_ZERO 1            /* the record holds code 0 ('=') */
#define HAP_OTHER 2           /* the record holds a symbol that is not exactly one of A, C, G, T */
// start / clen (round 6): the signature's start and the length of its contig, so that a pair's views come from its two records alone - k_edit_prep is bound by
// its chain of dependent loads, and work item -> start / contig columns -> contig offsets -> records was three links of it (now: work item -> records)
struct HapRec { unsigned long long word_off; int left, len, right, flags; int start, clen; };
// a haplotype as the pair kernels see it: `len` symbols starting `off` symbols into record words w[]
struct HapView { unsigned long long word_off; int off, len, flags; };

__device__ __forceinline__ int fetch_len(long long clen, long long a, long long b) {       // length of FastaFile.fetch(a, b) as hap_fetch clips it
    if (a < 0) a = 0;
    if (b < 0) b = 0;
    if (b > clen) b = clen;
    return a >= b ? 0 : (int)(b - a);
}

// where the two strings of a work item come from
struct PairSource {
    int plain;                       // 1: codes + a_off/b_off ; 0: signature pairs
    long long n_pairs;
    const uint8_t* codes; const int64_t* a_off; const int64_t* b_off;
    const EditWork* work; ClusterIn in; const int64_t* g_off; const uint8_t* g_codes;
    long long radius;                // R
    const HapRec* rec;
    __device__ __forceinline__ long long n_records() const { return plain ? 2 * n_pairs : in.n; }
    // the string record r stores, as a virtual concatenation of byte ranges
    __device__ __forceinline__ Hap record_hap(long long r) const {
        if (plain) {
            if (r < n_pairs) return plain_hap(codes + a_off[r], (int)(a_off[r + 1] - a_off[r]));
            r -= n_pairs;
            return plain_hap(codes + b_off[r], (int)(b_off[r + 1] - b_off[r]));
        }
        if (in.type[r] != SVX_INS) return plain_hap(g_codes, 0);
        const long long st = in.start[r];
        return make_hap(g_off, g_codes, in.contig[r], st, in.seq + in.seq_off[r], (int)(in.seq_off[r + 1] - in.seq_off[r]), st - radius, st + radius);
    }
    // compute_haplotype_edit_distance (src/svim/SVIM_clustering.py:32-45): window = min/max start -+ 100
    // shift = |start_a - start_b|: the haplotype of the later insertion carries that many reference bases in FRONT of its inserted
    // sequence which the other one carries BEHIND it, i.e. the alignment leaves the main diagonal by `shift` whatever the sequences are
    // shift > 0: A is the earlier insertion (A's inserted sequence lines up with B's `shift` symbols further right in B), < 0: B is
    __device__ __forceinline__ void views(long long w, HapView& A, HapView& B, int& shift) const {
        shift = 0;
        if (plain) {
            const HapRec ra = rec[w], rb = rec[n_pairs + w];
            A.word_off = ra.word_off; A.off = 0; A.len = ra.len; A.flags = ra.flags;
            B.word_off = rb.word_off; B.off = 0; B.len = rb.len; B.flags = rb.flags;
        } else {
            const EditWork wk = work[w];
            const HapRec ra = rec[wk.a], rb = rec[wk.b];
            const long long s1 = ra.start, s2 = rb.start;
            const long long ws = (s1 < s2 ? s1 : s2) - 100, we = (s1 > s2 ? s1 : s2) + 100;
            shift = (int)(s2 - s1);
            const long long l1 = ra.clen, l2 = rb.clen;
            const int la = fetch_len(l1, ws, s1), ra_ = fetch_len(l1, s1, we);
            const int lb = fetch_len(l2, ws, s2), rb_ = fetch_len(l2, s2, we);
            A.word_off = ra.word_off; A.off = ra.left - la; A.len = la + ra.len + ra_; A.flags = ra.flags;
            B.word_off = rb.word_off; B.off = rb.left - lb; B.len = lb + rb.len + rb_; B.flags = rb.flags;
        }
    }
    __device__ __forceinline__ long long slot(long long w) const { return plain ? w : work[w].slot; }
};

// per-pair descriptor after trimming
struct PairDesc {
    unsigned long long pat;     // packed-store word that holds the first symbol of the shorter core
    unsigned long long txt;     // same for the longer core
    int m, n;                   // core lengths, m <= n
    int ub;                     // upper bound of the distance known so far
    int cls;                    // bits 0..7 next class to try, bit 8 CLS_GENERIC, bit 9 CLS_ZERO, bits 12..14 / 16..18 nibble of the first pattern / text symbol,
                                // bits 20..30 the pair's position shift (clamped to 2047); -1 = answered by k_edit_prep
};
#define CLS_SHIFT(c) (((c) >> 20) & 2047)
#define CLS_PAT_SH(c) ((((c) >> 12) & 7) << 2)
#define CLS_TXT_SH(c) ((((c) >> 16) & 7) << 2)

// Band classes 0..NBAND-1: window of 32 * BAND_WORDS diagonals (0..2 sliding window, 3.. staircase window), tried in rounds.
#define NBAND 9
__host__ __device__ __forceinline__ int band_words(int b) { return b == 0 ? 1 : (b == 1 ? 3 : (b == 2 ? 4 : 2 * b)); }     // 1 | 3 4 6 8 10 12 14 16
// classes below FIRST_STAIR_CLS: sliding window (d_edit_band), from it on: staircase window (d_edit_stair).  The 4-word class was a sliding window
// until round 4: four v_alignbit per word and column make its word-column 54 issue cycles against 38, and it cannot narrow - as a staircase class
// (46 diagonals of slack instead of 14, so the pairs with 82 < needed window <= 114 start with 6 words) the step went from 21.55 to 20.4 ms
// (profiles/r04_edit_stair4_ab.txt)
// Round 6: the 2-word sliding class too - a v_alignbit costs ~8.8 cycles inside the update (myers_column.hpp), four of them per word and column make the sliding
// word-column ~2.3 x the staircase one; as a 3-word staircase window (96 rows: the same 50 diagonals of need, 46 of slack) the class pays 3 x 34 instead of 2 x ~90
// cycles per column.  Only the 1-word class (32 diagonals, patterns of a few dozen rows) still slides.
#define FIRST_STAIR_CLS 1
#define CLS_FULL 9            /* systolic full matrix, one wave per pair */
#define CLS_LANE0 10          /* 10..14: whole pattern (<= 32<<k rows) in one lane, full matrix, never fails */
#define CLS_WIDE0 15          /* 15..18: full matrix, 2/4/8/16 lanes per pair with 512 rows each (m <= 1024 / 2048 / 4096 / 8192) */
#define CLS_WIDE12 19         /* 19..22: the same with 384 rows (12 words) per lane (m <= 768 / 1536 / 3072 / 6144): halves the rows a pair pads up to */
#define CLS_WIDE14 23         /* 23..26: 14 words per lane (m <= 896 / 1792 / 3584 / 7168) */
#define CLS_WIDE10 27         /* 27..30: 10 words per lane (m <= 640 / 1280 / 2560 / 5120): with the four widths a pair pads up to 1/8 of its rows at most */
#define N_CLASSES 31
#define MIN_MARGIN 16
// A pair whose two cores hold only A/C/G/T (BAM codes 1,2,4,8) runs the 2-bit-plane kernels (P = 2); anything else (N, IUPAC codes,
// the '=' filler) the generic 4-plane ones (P = 4).  PairDesc.cls bit 8 carries that flag; the sort class is flag*32 + class.
#define CLS_GENERIC 0x100
#define CLS_ZERO 0x200           /* a record of the pair holds code 0 ('='): full-matrix classes only (the band kernels use 0 as the never-matching filler) */
#define GENERIC_BASE 32
#define N_SORT_CLASSES 64
__device__ __forceinline__ unsigned long long sort_class(int cls_with_flag) {
    return (unsigned long long)(((cls_with_flag & CLS_GENERIC) ? GENERIC_BASE : 0) + (cls_with_flag & 0xff));
}
// symbol as the P-plane kernels see it: P = 4 the BAM code itself, P = 2 A,C,G,T -> 0,1,2,3
template <int P> __device__ __forceinline__ uint32_t sym(uint32_t c) {
    if (P == 4) return c;
    return (((c >> 1) | (c >> 3)) & 1u) | ((((c >> 2) | (c >> 3)) & 1u) << 1);
}

// secondary sort key: pairs that share a wave should cost the same (full-matrix class: rows first, then text length)
__device__ __forceinline__ unsigned long long work_key(int cls, int m, int n);
// class of a pair that needs the full matrix
__device__ __forceinline__ int full_class_for(int m);
__device__ __forceinline__ int lane_class_for(int m) { return m <= 32 ? 0 : m <= 64 ? 1 : m <= 128 ? 2 : m <= 256 ? 3 : 4; }

__device__ __forceinline__ int band_class_for(int need_w) {      // smallest class whose window keeps the needed margin
    // class 0 (sliding window, d_edit_band): 14 diagonals are lost to the alignment of dmax to 7 (mod 8);
    // classes 1.. (staircase window, d_edit_stair): the window stands still for 32 columns, which costs 32 more diagonals
    for (int b = 0; b < NBAND; b++)
        if (need_w + (b < FIRST_STAIR_CLS ? 14 : 46) <= 32 * band_words(b)) return b;
    return CLS_FULL;
}
// first band class with at least `words` state words (CLS_FULL if none)
__device__ __forceinline__ int band_class_with_words(int words) {
    for (int b = 0; b < NBAND; b++) if (band_words(b) >= words) return b;
    return CLS_FULL;
}
// window that guarantees exactness when the true distance is <= ub
__device__ __forceinline__ int need_window(int m, int n, int ub) {
    int x = (ub - (n - m)) / 2;
    if (x < 0) x = 0;
    return (n - m) + 2 * x + 1;
}

// The sort key of the work list is 24 bits = three radix passes (round 6; 40 bits = five passes before: 0.11 ms of the window): 7 bits of sort class
// (SORT_ANSWERED = behind every class) over 17 bits of cost order.  The order inside a class only decides which pairs share a wave - results are written by slot.
#define SORT_COST_BITS 17
#define SORT_ANSWERED 127ull
__device__ __forceinline__ unsigned long long work_key(int cls, int m, int n) {
    // descending cost inside a class: the longest-running waves are dispatched first, the short ones fill the tail
    const unsigned long long top = (1ull << SORT_COST_BITS) - 1;
    unsigned long long k = (unsigned long long)n > top ? top : (unsigned long long)n;
    if (cls == CLS_FULL) {                                   // systolic: rows decide the lane count - 9 bits of rows (units of 64) over 8 bits of text length (units of 256)
        const unsigned long long mm = (unsigned long long)(m >> 6) > 511 ? 511 : (unsigned long long)(m >> 6), nn = (unsigned long long)(n >> 8) > 255 ? 255 : (unsigned long long)(n >> 8);
        k = (mm << 8) | nn;
    }
    return top - k;
}
__device__ __forceinline__ unsigned long long sort_key_of(unsigned long long sort_cls, unsigned long long cost_key) { return (sort_cls << SORT_COST_BITS) | cost_key; }

__device__ __forceinline__ int full_class_for(int m) {
    if (m <= 512) return CLS_LANE0 + lane_class_for(m);
#pragma unroll
    for (int k = 0; k < 4; k++) {
#ifndef SVX_NO_WIDE1014
        if (m <= (640 << k)) return CLS_WIDE10 + k;
#endif
        if (m <= (768 << k)) return CLS_WIDE12 + k;
#ifndef SVX_NO_WIDE1014
        if (m <= (896 << k)) return CLS_WIDE14 + k;
#endif
        if (m <= (1024 << k)) return CLS_WIDE0 + k;
    }
    return CLS_FULL;
}

#include "myers_column.hpp"

// Work accounting for the roofline of the edit kernels (bench.py `roofline_edit`): 32-bit word-columns a wave EXECUTES (issued: every lane
// pays for the longest text of its wave and for padded / idle state words) and the ones its pairs need (useful).  One atomic pair per
// wave at its very end, 32 shards; wc == nullptr switches it off.
#define WC_SHARDS 32
__device__ __forceinline__ void wc_account(unsigned long long* wc, long long issued_wave, long long useful_lane) {
    if (!wc) return;
    const long long useful = wave_sum_i64(useful_lane);
    if (lane_id() == 0) {
        unsigned long long* w = wc + 2 * (blockIdx.x & (WC_SHARDS - 1));
        atomicAdd(w, (unsigned long long)issued_wave);
        atomicAdd(w + 1, (unsigned long long)useful);
    }
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}

// ---- 1. packed store ---------------------------------------------------------------------------------------------
__global__ void k_pair_span(long long n_work, PairSource src, unsigned long long* span) {           // max |start_a - start_b| over the work list
    long long d = 0;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < n_work; w += (long long)gridDim.x * blockDim.x) {
        const EditWork wk = src.work[w];
        long long x = (long long)src.in.start[wk.a] - (long long)src.in.start[wk.b];
        if (x < 0) x = -x;
        d = x > d ? x : d;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const long long v = __shfl_xor(d, o, 64); d = v > d ? v : d; }
    if (lane_id() == 0 && d > 0) atomicMax(span + 16 + (blockIdx.x & 15), (unsigned long long)d);          // 16 shards
}

__global__ void k_hap_words(long long n_rec, PairSource src, int64_t* words) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rec) return;
    if (r == n_rec) { words[r] = 0; return; }
    const Hap h = src.record_hap(r);
    words[r] = h.len ? (h.len + 7) / 8 + 2 : 0;          // two pad words: word(idx) may read one word past the last symbol
}

// one wave per record: 4-bit pack (8 symbols per lane and step), note the alphabet
__global__ __launch_bounds__(256) void k_hap_pack(long long n_rec, PairSource src, const int64_t* word_off, uint32_t* packed, HapRec* rec) {
    const long long r = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (r >= n_rec) return;
    const int lane = lane_id();
    const Hap h = src.record_hap(r);
    const unsigned long long base = (unsigned long long)word_off[r] + 1ull;          // word 0 of the store is a pad
    uint32_t* out = packed + base;
    int zero = 0, other = 0;
    for (int i0 = lane * 8; h.len > 0 && i0 < h.len + 16; i0 += 512) {                 // + 16: the pad words are zeroed; an empty record owns no words
        const uint32_t wd = i0 < h.len ? h.pack8(i0, h.len) : 0u;
        out[i0 >> 3] = wd;
        const int v = h.len - i0 >= 8 ? 8 : (h.len - i0 < 0 ? 0 : h.len - i0);
        const uint32_t vm = v >= 8 ? 0xffffffffu : ((1u << (4 * v)) - 1u);
        const uint32_t nz = (wd | (wd >> 1) | (wd >> 2) | (wd >> 3)) & 0x11111111u;
        zero |= (__popc(nz) < v);
        const uint32_t ones = (wd & 0x11111111u) + ((wd >> 1) & 0x11111111u) + ((wd >> 2) & 0x11111111u) + ((wd >> 3) & 0x11111111u);
        other |= ((ones ^ 0x11111111u) & vm) != 0u;                                   // a symbol that is not exactly one of A,C,G,T
    }
    zero = __any(zero); other = __any(other);
    if (lane == 0) {
        HapRec hr; hr.word_off = base; hr.left = h.n0; hr.len = h.n1; hr.right = h.n2; hr.flags = (zero ? HAP_ZERO : 0) | (other ? HAP_OTHER : 0);
        hr.start = 0; hr.clen = 0;
        if (src.plain) { hr.left = 0; hr.len = h.len; hr.right = 0; }
        else {
            const int ctg = src.in.contig[r];
            const long long cl = src.g_off[ctg + 1] - src.g_off[ctg];
            hr.start = src.in.start[r]; hr.clen = cl > 0x7fffffffll ? 0x7fffffff : (int)cl;      // (BAM positions are 32-bit: no contig is longer)
        }
        rec[r] = hr;
    }
}

#ifndef PREP_LANES
#define PREP_LANES 16
#endif
// ---- 2. trim + classify (PREP_LANES lanes per pair, reads only the packed store) ---------------------------------------------------
// G lanes per pair (64 / G pairs per wave): the kernel is bound by the latency of its dependent loads, so what counts is the number of
// pairs in flight, not the symbols compared per step
template <int G>
__global__ __launch_bounds__(256) void k_edit_prep(long long n_work, PairSource src, const uint32_t* packed, PairDesc* desc,
                                                   uint64_t* sort_key, uint32_t* sort_val, int32_t* ed, unsigned long long* cells, int shift_bounds) {
    const int wave_lane = lane_id();
    const int sg = wave_lane / G, lane = wave_lane % G;                   // sub-group of the wave / lane inside it
    const unsigned long long sg_mask = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (sg * G);
    // (Round 6, measured and dropped: an XCD-aware block order - every XCD takes a contiguous eighth of the work list, so that the pairs of a partition, which share
    // their packed haplotypes, meet in ONE L2 instead of all eight - 1.70 ms either way, the step 0.15 ms slower: profiles/r06_edit_class1_stair_prep_xcd_ab.txt)
    const long long w = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / G) + sg;
    if (w >= n_work) return;
    HapView A, B;
    int sshift;
    src.views(w, A, B, sshift);
    const int shift = sshift < 0 ? -sshift : sshift;
    const uint32_t* wa = packed + A.word_off; const uint32_t* wb = packed + B.word_off;
    const int la = A.len, lb = B.len;
    const int mn = la < lb ? la : lb;
    // common prefix / suffix, 8 symbols per lane and step.  The first trips of the two walks are loaded together (round 6: the suffix walk's addresses do
    // not depend on the prefix, only its symbol count does - one round trip less in the chain of dependent loads)
    uint32_t xp0 = 0, xs0 = 0;
    if (mn - lane * 8 > 0) {
        const uint32_t pa = fetch8(wa, A.off + lane * 8), pb = fetch8(wb, B.off + lane * 8);
        const uint32_t sa = fetch8(wa, A.off + la - lane * 8 - 8), sb = fetch8(wb, B.off + lb - lane * 8 - 8);
        xp0 = pa ^ pb; xs0 = sa ^ sb;
    }
    int pre = mn;
    for (int base = 0; base < mn; base += 8 * G) {
        const int i0 = base + lane * 8;
        const int cnt = mn - i0 >= 8 ? 8 : (mn - i0 < 0 ? 0 : mn - i0);
        uint32_t x = 0;
        if (base == 0) x = cnt > 0 ? xp0 : 0u;
        else if (cnt > 0) x = fetch8(wa, A.off + i0) ^ fetch8(wb, B.off + i0);
        const uint32_t nz = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;
        int k = nz ? (__ffs((int)nz) - 1) >> 2 : 8;                       // first differing symbol of the chunk
        if (k > cnt) k = cnt;
        const unsigned long long stop = __ballot(k < 8) & sg_mask;         // a difference, or the shorter string ends here
        if (stop) { const int f = __ffsll((long long)stop) - 1; pre = base + (f - sg * G) * 8 + __shfl(k, f, 64); break; }
    }
    const int lim = mn - pre;
    int suf = lim;
    for (int base = 0; base < lim; base += 8 * G) {
        const int i0 = base + lane * 8;
        const int cnt = lim - i0 >= 8 ? 8 : (lim - i0 < 0 ? 0 : lim - i0);
        uint32_t x = 0;
        if (base == 0) x = cnt > 0 ? xs0 : 0u;
        else if (cnt > 0) x = fetch8(wa, A.off + la - i0 - 8) ^ fetch8(wb, B.off + lb - i0 - 8);
        const uint32_t nz = (x | (x << 1) | (x << 2) | (x << 3)) & 0x88888888u;
        int k = nz ? __clz((int)nz) >> 2 : 8;                              // matching symbols counted from the end of the chunk
        if (k > cnt) k = cnt;
        const unsigned long long stop = __ballot(k < 8) & sg_mask;
        if (stop) { const int f = __ffsll((long long)stop) - 1; suf = base + (f - sg * G) * 8 + __shfl(k, f, 64); break; }
    }
    const int ca = la - pre - suf, cb = lb - pre - suf;
    PairDesc pd;
    const bool a_short = ca <= cb;
    const HapView& P = a_short ? A : B;
    const HapView& T = a_short ? B : A;
    pd.m = a_short ? ca : cb; pd.n = a_short ? cb : ca;
    const int p0 = P.off + pre, t0 = T.off + pre;                          // first core symbols inside their records
    pd.pat = P.word_off + (unsigned long long)(p0 >> 3);
    pd.txt = T.word_off + (unsigned long long)(t0 >> 3);
    const int sh_bits = ((p0 & 7) << 12) | ((t0 & 7) << 16) | ((shift > 2047 ? 2047 : shift) << 20);
    if (pd.m == 0) {                                   // one core is empty: the distance is the other's length
        if (lane == 0) { ed[src.slot(w)] = pd.n; pd.ub = pd.n; pd.cls = -1; desc[w] = pd; sort_key[w] = sort_key_of(SORT_ANSWERED, 0); sort_val[w] = (uint32_t)w; }
        return;
    }
    // Upper bounds from trivial alignments: substitutions only, the pattern pushed `a` and the text `b` symbols to the right
    //   cost = a + b + mismatches over the overlap + what is left of either core behind it.
    // (0, 0) is the left-justified one.  Two insertions at DIFFERENT positions carry `shift` reference bases on opposite sides of their inserted
    // sequences (PairSource::views): their inserted sequences line up at (0, shift) or (shift, 0), whichever core belongs to the earlier
    // insertion - tried first for such a pair, (0, 0) only when that says little.  A bound is only worth its pass over the cores when it is SMALL: after the first 8 G
    // symbols an alignment with more than 25 % mismatches is given up (three quarters of the positions of unrelated or misaligned sequences
    // mismatch from the first symbols on).  A tight bound does two things: the pair starts in the narrowest band that certifies it, and the
    // staircase window narrows against the bound instead of against the window's own capacity (d_edit_stair).
    const uint32_t* wp = packed + P.word_off; const uint32_t* wt = packed + T.word_off;
    // (Round 6, measured and dropped - none of them moves the kernel's 1.7 ms: four chunks per lane and trip with all eight loads issued up front (1.94-2.04 ms: the
    // lanes of a last trip load for nothing); 32 symbols per lane and trip - five words of either string, four funnel shifts, ~45 instructions per 32 symbols instead
    // of ~25 per 8 (1.76 ms); an XCD-aware block order (1.70 ms).  The bound pass is not what the kernel waits for; the chain of dependent loads in front of it -
    // work item -> signature columns -> contig offsets -> record headers -> first words - is the suspect, not followed up.)
    auto bound_at = [&](const int a, const int b) -> int {                  // every lane of the sub-group returns the same value; m + n = useless
        const int L = (pd.m - a) < (pd.n - b) ? (pd.m - a) : (pd.n - b);
        if (L <= 0) return pd.m + pd.n;
        int ham_l = 0;
        bool useless = false;
        for (int base = 0; base < L; base += 8 * G) {
            const int i0 = base + lane * 8;
            if (i0 < L) {
                const int v = L - i0 >= 8 ? 8 : L - i0;
                const uint32_t vmask = v >= 8 ? 0xffffffffu : ((1u << (4 * v)) - 1u);
                const uint32_t x = (fetch8(wp, p0 + a + i0) ^ fetch8(wt, t0 + b + i0)) & vmask;
                ham_l += __popc((x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u);
            }
            if (base == 0 && L > 8 * G) {
                int first = ham_l;
#pragma unroll
                for (int o = G / 2; o >= 1; o >>= 1) first += __shfl_xor(first, o, 64);
                if (first > 2 * G) { useless = true; break; }
            }
        }
#pragma unroll
        for (int o = G / 2; o >= 1; o >>= 1) ham_l += __shfl_xor(ham_l, o, 64);          // sum over the sub-group
        if (useless) return pd.m + pd.n;
        return a + b + ham_l + (pd.m - a - L) + (pd.n - b - L);
    };
    // the core of the EARLIER insertion is the one whose inserted sequence comes first: its symbols line up with the other core's `shift` further right
    const bool pat_earlier = (sshift > 0) == a_short;
    int ub = pd.n;                                                          // substitute the shorter core, insert the rest
    if (shift_bounds && shift > 0 && shift < (pat_earlier ? pd.n : pd.m)) ub = pat_earlier ? bound_at(0, shift) : bound_at(shift, 0);
    if (ub > pd.n) ub = pd.n;
    if (4 * ub > pd.n) { const int u = bound_at(0, 0); ub = u < ub ? u : ub; }         // (also the only one tried for insertions at the same position)
    const int zero = (A.flags | B.flags) & HAP_ZERO, other = (A.flags | B.flags) & HAP_OTHER;      // of the whole records: conservative
    if (lane == 0) {
        pd.ub = ub;
        pd.cls = (other ? CLS_GENERIC : 0) | (zero ? CLS_ZERO : 0) | sh_bits;       // the class itself: k_edit_classify
        desc[w] = pd;
        if (cells) atomicAdd(cells + (w & 1023), (unsigned long long)pd.m * (unsigned long long)pd.n);      // 1024 shards: no same-address pile-up
    }
}

// first class of every pair + its sort key (one thread per pair)
// classes whose round-0 pairs run in row blocks (SVX_EDIT_BLOCKED): the 2- and 4-lane families (<= 4 blocks: the blocks of a pair are serial launches); words per block
__host__ __device__ __forceinline__ int blocked_words(int cls, int kmax) {
    if (cls >= CLS_WIDE0 && cls <= CLS_WIDE0 + kmax) return 16;
    if (cls >= CLS_WIDE12 && cls <= CLS_WIDE12 + kmax) return 12;
    if (cls >= CLS_WIDE14 && cls <= CLS_WIDE14 + kmax) return 14;
    if (cls >= CLS_WIDE10 && cls <= CLS_WIDE10 + kmax) return 10;
    return 0;
}
__global__ void k_edit_classify(long long n_work, PairDesc* desc, uint64_t* sort_key, uint32_t* sort_val, int force_full, unsigned long long* guess_word, int few_pairs_flags) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int few_pairs = few_pairs_flags & 1, blocked_kmax = (few_pairs_flags >> 1) - 1;          // -1: no blocked route
    if (w >= n_work) return;
    const PairDesc pd = desc[w];
    if (pd.cls == -1) return;                               // empty core: answered by k_edit_prep, key already written
    int cls;
    if (force_full) cls = CLS_FULL;
    else if (pd.cls & CLS_ZERO) cls = full_class_for(pd.m);
    else {
        // `guaranteed` cannot fail (the trivial alignments bound the distance); when that bound is useless (position jitter
        // shifts the two cores against each other) start from a band sized for guess_frac * m differences and widen on failure.
        const int guaranteed = band_class_for(need_window(pd.m, pd.n, pd.ub));
        int guess = 2 * MIN_MARGIN;
        const float guess_frac = __uint_as_float((uint32_t)guess_word[0]);          // k_edit_guess / the host's fallback (cnt[2])
        const int by_frac = (int)ceilf(guess_frac * (float)pd.m);
        if (by_frac > guess) guess = by_frac;
        guess += 2 * CLS_SHIFT(pd.cls);                 // what the position shift alone costs (see PairSource::views)
        int spec = band_class_for((pd.n - pd.m) + guess + 1);
        // widest band before giving up on banding - unless the position shift alone (which costs 2 * shift whatever the sequences are)
        // already exceeds it: such a pair is either unrelated or out of every band's reach, the attempt would only delay its full matrix
        if (spec == CLS_FULL && band_class_for((pd.n - pd.m) + 2 * CLS_SHIFT(pd.cls) + 2 * MIN_MARGIN + 1) != CLS_FULL) spec = NBAND - 1;
        cls = guaranteed <= spec ? guaranteed : spec;
        // short patterns: the whole column fits one lane (k_edit_lane), nothing to speculate about
        if (cls > 0 && cls < NBAND && pd.m <= 512 && (1 << lane_class_for(pd.m)) <= band_words(cls)) cls = CLS_LANE0 + lane_class_for(pd.m);
        else if (cls == CLS_FULL) cls = full_class_for(pd.m);
    }
    // A call with few pairs is bound by the LATENCY of its longest pair, not by throughput: a lane-per-pair kernel walks a pair's columns in one
    // lane (4000 columns x 12 band words x 40 cycles = 0.9 ms; 1700 columns x 16 words of a 400-row pattern = 0.45 ms), the 64-lane full-matrix forms
    // (d_edit_wide<64, Q>, 2-4 words per lane, chosen by the launch when few waves are in flight) finish the same pair in a fraction of that while
    // most of the chip idles either way.  Short patterns and narrow bands of short pairs stay where they are (one or two words per column already).
    if (few_pairs && !force_full && pd.m > 64 && !(cls < NBAND && pd.m <= 512))
        cls = pd.m <= 4096 ? CLS_WIDE0 + 2 : (pd.m <= 6144 ? CLS_WIDE12 + 3 : (pd.m <= 8192 ? CLS_WIDE0 + 3 : CLS_FULL));
    const int flagged = cls | (pd.cls & ~0xff);
    desc[w].cls = flagged;
    int cost_n = pd.n;
    const int bq = blocked_kmax >= 0 ? blocked_words(cls, blocked_kmax) : 0;
    if (bq) {
        // a row block walks 32 Q + (length gap) + 2 x columns (d_edit_blocked): pairs that share a wave should agree on THAT; and the longest text of the
        // blocked pairs sizes their boundary words (guess_word[1], read by the host with the class bounds)
        int ub = pd.ub; if (ub > pd.m + pd.n) ub = pd.m + pd.n; if (ub < pd.n - pd.m) ub = pd.n - pd.m;
        const long long walk = 32ll * bq + (pd.n - pd.m) + 2ll * ((ub - (pd.n - pd.m)) / 2 + 1);
        cost_n = walk < pd.n ? (int)walk : pd.n;
        if ((unsigned long long)pd.n > __hip_atomic_load(guess_word + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(guess_word + 1, (unsigned long long)pd.n);
    }
    sort_key[w] = sort_key_of(sort_class(flagged), work_key(cls, pd.m, cost_n));
    sort_val[w] = (uint32_t)w;
}

// a band kernel could not certify its result: d (a valid alignment cost, hence an upper bound) picks the next class
__device__ __forceinline__ void band_retry(const PairDesc& pd, uint32_t widx, int m, int n, int d, PairDesc* desc, unsigned long long* fail_cnt,
                                           uint32_t* fail_lists, long long fail_cap) {
    const int ub = d < pd.ub ? d : pd.ub;
    const int cur = pd.cls & 0xff;
    int cls = band_class_for(need_window(m, n, ub));   // cannot fail, but d from a too-narrow band can be a gross over-estimate:
    const int lo = band_class_with_words(2 * band_words(cur)), hi = band_class_with_words(4 * band_words(cur));
    if (cls > hi) cls = hi;                            // widen geometrically (at most x4) instead
    if (cls < lo) cls = lo;                            // and at least x2: a retry at nearly the same width mostly fails again
    if (cls >= NBAND) cls = CLS_FULL;
    if (cls < NBAND && m <= 512 && (1 << lane_class_for(m)) <= band_words(cls)) cls = CLS_LANE0 + lane_class_for(m);
    else if (cls == CLS_FULL) cls = full_class_for(m);
    const int flagged = cls | (pd.cls & ~0xff);          // keeps the alphabet flag and the nibble offsets
    desc[widx].ub = ub; desc[widx].cls = flagged;
    // retry list of the new class (the lists of a round are consumed as they are: no re-sort between rounds)
    const unsigned long long sc = sort_class(flagged);
    const unsigned long long i = atomicAdd(fail_cnt + sc, 1ull);
    fail_lists[sc * (unsigned long long)fail_cap + i] = widx;
}

// ---- 3. banded lane-per-pair kernel -----------------------------------------------------------------------------
// Window of W = 32*Q bits; bit b of column j <-> row (j - dmax) + b.  dmax = 7 (mod 8) so that the row entering at the
// bottom of the window and the text symbol of the column sit at the same nibble phase of their packed words.
// returns the cost d of the best path inside the window (an upper bound of the distance) and the window's margin: d is the exact
// distance iff 0 <= floor((d - (n-m)) / 2) <= margin.  Every lane of the wave must call (uniform trip count); dead lanes pass live = false.
template <int Q, int P>
__device__ __forceinline__ int band_core(const bool live, const PairDesc& pd, const uint32_t* scratch, int& margin_out) {
    const int W = 32 * Q;
    const int m = pd.m, n = live ? pd.n : 0;
    // largest dmax <= (n-m) + floor((W-1-(n-m))/2) with dmax = 7 (mod 8)
    int a0 = (W - 1 - (n - m)) / 2;
    int dmax = (n - m) + a0;
    dmax -= ((dmax - 7) & 7);
    const int margin = dmax - (n - m);                 // top margin; bottom margin W-1-dmax >= margin
    const int a_bot = W - 1 - dmax;                    // multiple of 8
    const Packed pat{scratch + pd.pat, CLS_PAT_SH(pd.cls)}, txt{scratch + pd.txt, CLS_TXT_SH(pd.cls)};
    uint32_t pv[Q], mv[Q], pl[P][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int lo = 32 * q, nvirt = dmax + 1;        // rows <= 0 : vertical delta -1
        uint32_t mlow;
        if (nvirt >= lo + 32) mlow = 0xffffffffu; else if (nvirt <= lo) mlow = 0u; else mlow = (1u << (nvirt - lo)) - 1u;
        mv[q] = mlow; pv[q] = ~mlow;
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = 0u;
    }
    const int pat_words = (m + 7) >> 3;
    // pre-roll: rows 1..a_bot enter the window (a_bot/8 whole words)
    for (int wi = 0; wi < (a_bot >> 3); wi++) {
        const uint32_t word = (live && wi < pat_words) ? pat.word(wi) : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int row = wi * 8 + k + 1;
            const uint32_t c = (row <= m) ? sym<P>((word >> (4 * k)) & 15u) : 0u;
#pragma unroll
            for (int b = 0; b < P; b++) {
#pragma unroll
                for (int q = 0; q < Q - 1; q++) pl[b][q] = __builtin_amdgcn_alignbit(pl[b][q + 1], pl[b][q], 1);
                pl[b][Q - 1] = (pl[b][Q - 1] >> 1) | (((c >> b) & 1u) << 31);
            }
        }
    }
    int S = dmax;
    // uniform trip count: the longest text in the wave
    int nmax = n;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    const int txt_words = (n + 7) >> 3;
    const int pbase = a_bot >> 3;                        // pattern word holding row j + a_bot for j = 8*jb+1..
    uint32_t tw_next = (live && txt_words > 0) ? txt.word(0) : 0u;
    uint32_t pw_next = (live && pbase < pat_words) ? pat.word(pbase) : 0u;
    for (int jb = 0; jb * 8 < nmax; jb++) {
        const uint32_t tw = tw_next, pw = pw_next;
        tw_next = (live && jb + 1 < txt_words) ? txt.word(jb + 1) : 0u;
        pw_next = (live && pbase + jb + 1 < pat_words) ? pat.word(pbase + jb + 1) : 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = jb * 8 + k + 1;
            if (j <= n) {
                const uint32_t c = sym<P>((tw >> (4 * k)) & 15u);
                const uint32_t pc = (j + a_bot <= m) ? sym<P>((pw >> (4 * k)) & 15u) : 0u;
                // slide the window one row down
#pragma unroll
                for (int q = 0; q < Q - 1; q++) { pv[q] = __builtin_amdgcn_alignbit(pv[q + 1], pv[q], 1); mv[q] = __builtin_amdgcn_alignbit(mv[q + 1], mv[q], 1); }
                pv[Q - 1] = (pv[Q - 1] >> 1) | 0x80000000u; mv[Q - 1] >>= 1;
                uint32_t nk[P];
#pragma unroll
                for (int b = 0; b < P; b++) {
#pragma unroll
                    for (int q = 0; q < Q - 1; q++) pl[b][q] = __builtin_amdgcn_alignbit(pl[b][q + 1], pl[b][q], 1);
                    pl[b][Q - 1] = (pl[b][Q - 1] >> 1) | (((pc >> b) & 1u) << 31);
                    nk[b] = ((c >> b) & 1u) - 1u;
                }
                S += (int)(pv[0] & 1u) - (int)(mv[0] & 1u);
                uint32_t carry = 0, ph_in = 1u, mh_in = 0u;
#pragma unroll
                for (int q = 0; q < Q; q++) {
                    uint32_t eq = pl[0][q] ^ nk[0];
#pragma unroll
                    for (int b = 1; b < P; b++) eq &= pl[b][q] ^ nk[b];
                    const uint32_t PV = pv[q], MV = mv[q];
                    const uint32_t xv = eq | MV;
                    unsigned carry_out;
                    const uint32_t sum = __builtin_addc(eq & PV, PV, carry, &carry_out);      // v_addc_co_u32: the carry stays in an SGPR pair
                    carry = carry_out;
                    const uint32_t xh = (sum ^ PV) | eq;
                    uint32_t ph = MV | ~(xh | PV);
                    uint32_t mh = PV & xh;
                    if (q == 0) S += (int)(ph & 1u) - (int)(mh & 1u);
                    const uint32_t ph_out = ph >> 31, mh_out = mh >> 31;
                    ph = (ph << 1) | ph_in; mh = (mh << 1) | mh_in;
                    ph_in = ph_out; mh_in = mh_out;
                    pv[q] = mh | ~(xv | ph);
                    mv[q] = ph & xv;
                }
            }
        }
    }
    margin_out = margin;
    // D[m][n] = D[top][n] + vertical deltas of rows top+1..m  (bits 1..margin)
    int d = S;
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int lo = 32 * q;                           // bits lo..lo+31 ; wanted bits 1..margin
        int hi_bit = margin - lo;                        // number of wanted bits in this word counted from bit 0 of the word, inclusive of bit 'margin'
        uint32_t mask;
        if (hi_bit >= 31) mask = 0xffffffffu; else if (hi_bit < 0) mask = 0u; else mask = (2u << hi_bit) - 1u;
        if (q == 0) mask &= ~1u;
        d += __popc(pv[q] & mask) - __popc(mv[q] & mask);
    }
    return d;
}

template <int Q, int P>
__device__ __forceinline__ void d_edit_band(long long blk, long long count, const uint32_t* list, const uint32_t* scratch, PairDesc* desc,
                                                   const long long* slot_of, int32_t* ed, unsigned long long* fail_cnt, uint32_t* fail_lists,
                                                   long long fail_cap, unsigned long long* wc) {
    const long long t = blk * 256 + threadIdx.x;
    const bool live = t < count;
    uint32_t widx = 0;
    PairDesc pd; pd.m = 0; pd.n = 0; pd.pat = 0; pd.txt = 0; pd.ub = 0; pd.cls = 0;
    if (live) { widx = list[t]; pd = desc[widx]; }
    int margin;
    const int d = band_core<Q, P>(live, pd, scratch, margin);
    wc_account(wc, (long long)wave_max_i32(live ? pd.n : 0) * Q * 64, live ? (long long)pd.n * Q : 0);
    if (!live) return;
    const int m = pd.m, n = pd.n;
    const int x = (d - (n - m)) >> 1;                    // floor: d >= n-m always
    if (margin >= 0 && x >= 0 && x <= margin) ed[slot_of ? slot_of[widx] : (long long)widx] = d;
    else band_retry(pd, widx, m, n, d, desc, fail_cnt, fail_lists, fail_cap);
}

// ---- 3'. pilot: how far apart are this call's pairs? -------------------------------------------------------------------------
// A pair whose trivial alignments give no useful bound starts in a band sized for 2 * shift + guess_frac * (core length) differences
// beyond the length gap (k_edit_classify): the position shift of the two insertions is known, how much their SEQUENCES differ is
// not.  guess_frac is chosen per CALL from a strided sample of its own pairs: the 64-diagonal band kernel over the first <= 384
// symbols of both cores gives each sampled pair's cost, minus the 2 * shift that leaving and re-joining the main diagonal costs
// = its sequence divergence per symbol (beyond ~10 % only an upper bound, which is all the cost model needs), weighted by the core
// length like the work it stands for.  No state is carried from one call to the next; routing only, results never depend on it.
#define PILOT_MAX 16384
#define PILOT_PREFIX 384
__global__ __launch_bounds__(256) void k_edit_pilot(long long n_work, long long stride, const PairDesc* desc, const uint32_t* scratch, unsigned long long* hist) {
    __shared__ unsigned long long h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long w = ((long long)blockIdx.x * 256 + threadIdx.x) * stride;
    PairDesc pd; pd.m = 0; pd.n = 0; pd.pat = 0; pd.txt = 0; pd.ub = 0; pd.cls = 0;
    bool live = w < n_work;
    int full_m = 0;
    if (live) {
        pd = desc[w];
        full_m = pd.m;
        // '=' is the band kernel's filler symbol; short cores do not speculate; a large shift does not fit the pilot's band
        live = pd.cls != -1 && !(pd.cls & CLS_ZERO) && pd.m >= 128 && CLS_SHIFT(pd.cls) <= 12;
        const int L = pd.m < PILOT_PREFIX ? pd.m : PILOT_PREFIX;
        pd.m = L; pd.n = L;
    }
    int margin;
    const int d = band_core<2, 4>(live, pd, scratch, margin);
    if (live) {
        long long b = (long long)(d - 2 * CLS_SHIFT(pd.cls)) * 256 / pd.m;
        if (b < 0) b = 0;
        if (b > 255) b = 255;
        atomicAdd(&h[b], (unsigned long long)full_m);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(hist + threadIdx.x, h[threadIdx.x]);
}

// The speculation fraction from the pilot's histogram, on the device (round 6: the host used to fetch the 256 bins and wait for them - one stream
// synchronisation less in front of the rounds).  Same rule as guess_from_histogram below (kept for the profile output): the g with the least expected
// cost; thread b prices g = (b + 1) / 256, the first of the cheapest wins.  out[0] = the fraction (float bits), unchanged when the histogram is empty.
__global__ __launch_bounds__(128) void k_edit_guess(const unsigned long long* hist, unsigned long long* out) {
    __shared__ double cum[257];
    __shared__ double cost_of[128];
    if (threadIdx.x == 0) {
        double total = 0;
        cum[0] = 0;
        for (int b = 0; b < 256; b++) { total += (double)hist[b]; cum[b + 1] = total; }
    }
    __syncthreads();
    const double total = cum[256];
    const int b = (int)threadIdx.x;
    double cost = 1e300;
    if (b >= 3 && b < 128 && total > 0) {
        const double g = (b + 1) / 256.0;
        cost = g;
        for (double w = g; w < 1.0; w *= 2) {
            int q = (int)(w * 256.0); if (q > 256) q = 256;
            cost += (2 * w <= 0.5 ? 2 * w : 0.75) * (1.0 - cum[q] / total);
        }
    }
    cost_of[b] = cost;
    __syncthreads();
    if (threadIdx.x == 0 && total > 0) {
        int best = 31; double best_cost = 1e300;
        for (int k = 3; k < 128; k++) if (cost_of[k] < best_cost) { best_cost = cost_of[k]; best = k; }
        out[0] = (unsigned long long)__float_as_uint((float)((best + 2) / 256.0));
    }
}

// What the host wants to know when the rounds are over, in one post (svx_mail_*): [0] pairs too long for the register forms, [1] the speculation fraction
// (float bits), [2..5] word-columns issued / useful / of retry rounds / of band launches, summed over the shards of every (round, kind) counter block.
__global__ __launch_bounds__(256) void k_edit_tail(const unsigned long long* cnt, const unsigned long long* wc, int rounds, unsigned long long* out) {
    __shared__ unsigned long long acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long iss = 0, use = 0, retry = 0, band = 0;
    const int per = 2 * WC_SHARDS;
    for (int i = (int)threadIdx.x; i < rounds * 2 * WC_SHARDS; i += 256) {
        const int blockno = i / WC_SHARDS, sh = i % WC_SHARDS, r = blockno >> 1, k = blockno & 1;
        const unsigned long long a = wc[(size_t)blockno * per + 2 * sh], u = wc[(size_t)blockno * per + 2 * sh + 1];
        iss += a; use += u;
        if (r > 0) retry += a;
        if (k == 0) band += a;
    }
    atomicAdd(&acc[0], iss); atomicAdd(&acc[1], use); atomicAdd(&acc[2], retry); atomicAdd(&acc[3], band);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = cnt[1]; out[1] = cnt[2]; }
    if (threadIdx.x < 4) out[2 + threadIdx.x] = acc[threadIdx.x];
}

// ---- 3a. banded lane-per-pair kernel, staircase window (classes 1..8) ------------------------------------------------------
// Sliding the window one row per column costs four v_alignbit per state word and column - and v_alignbit is a half-rate
// instruction on gfx950 (4 cycles per wave64 against 2 for v_xor / v_bitop3; tools/micro/valu_ops.hip), 44 % of the sliding
// kernel's cycles.  Here the window of W = 32*Q rows stands still for a block of 32 columns and then drops by one whole word:
// the per-column work is the plain multi-word Myers/Hyyro update, the drop is 4*Q register moves plus one bit-plane word
// built from 4 packed pattern words.  Block kb (columns 32*kb+1 .. 32*kb+32) covers rows 32*kb-off .. 32*kb-off+W-1; the
// row above the window is assumed to step +1 per column and a word entering at the bottom to step +1 per row (both upper
// bounds, exact in the virtual rows <= 0).  A path of cost d leaves the corridor of diagonals [0, n-m] by at most
// x = floor((d-(n-m))/2) on either side, so the result is exact iff x <= min(W-off-33, off+1-(n-m)); off = 7 (mod 8) keeps
// the pattern words that enter aligned.
__device__ __forceinline__ uint32_t squeeze8(uint32_t x) {       // bits 0,4,..,28 -> bits 0..7
    x = (x | (x >> 3)) & 0x03030303u;
    x = (x | (x >> 6)) & 0x000f000fu;
    return (x | (x >> 12)) & 0xffu;
}
template <int P>
__device__ __forceinline__ void planes8(uint32_t w, uint32_t (&out)[P]) {       // one packed word (8 symbols) -> 8 bits of each plane
    if (P == 2) {
        out[0] = squeeze8(((w >> 1) | (w >> 3)) & 0x11111111u);
        out[1] = squeeze8(((w >> 2) | (w >> 3)) & 0x11111111u);
    } else {
#pragma unroll
        for (int b = 0; b < P; b++) out[b] = squeeze8((w >> b) & 0x11111111u);
    }
}

// bit-plane words of 32 rows given as 4 packed words.  Inlined: as a call it pins the callers' register arrays to the registers the calling
// convention preserves (the staircase kernel went from 147 to 264 VGPRs with the call in its drop).
template <int P>
__device__ __forceinline__ void planes32(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t (&out)[P]) {
    uint32_t e0[P], e1[P], e2[P], e3[P];
    planes8<P>(w0, e0); planes8<P>(w1, e1); planes8<P>(w2, e2); planes8<P>(w3, e3);
#pragma unroll
    for (int b = 0; b < P; b++) out[b] = e0[b] | (e1[b] << 8) | (e2[b] << 16) | (e3[b] << 24);
}

// The staircase window NARROWS while it runs (Ukkonen's cut-off, as edlib applies it to its block range): with kcap = the largest distance this
// attempt can certify (what the static window guarantees, or a smaller upper bound known beforehand), a cell whose value plus the
// remaining diagonal distance |(n - j) - (m - i)| exceeds kcap lies on no alignment of cost <= kcap.  At every drop (each 32 columns) the wave
// asks two questions about the column it has just finished (j1):
//   top:    would the last row r of the SECOND window word be such a cell?  (value = top + deltas of words 0 and 1, on diagonal j1 - r >= n - m).
//           Values never decrease along a diagonal and lose at most 1 per step towards the corridor, so every later cell on or above
//           that diagonal is such a cell too: both words leave instead of one.
//   bottom: is the last row of the second-to-last word such a cell (diagonal <= n - m)?  Then so is everything below it in the next 32
//           columns: no new word enters.  (Also when the window already reaches below row m.)
// Either answer must hold for every pair of the wave that still has columns (one __all each): the number of live words Q is wave-uniform and
// selects the unrolled column code.  A pair is answered iff its result is <= kcap - exactly the acceptance rule of the static window
// (d <= delta + 2 margin + 1), so narrowing changes what is computed, never what is accepted.  SVX_EDIT_NARROW=0 keeps the window static.
#define STAIR_QMAX 16            /* widest class; a launch without the two widest classes runs the kernel built for 12 words (more waves per SIMD) */
#ifndef STAIR_QMIN
#define STAIR_QMIN 3
#endif

template <int Q, int P, bool PRED, int QM>
__device__ __forceinline__ void stair_columns8(uint32_t (&pv)[QM], uint32_t (&mv)[QM], uint32_t (&pl)[P][QM], const uint32_t tword,
                                               const int j0, const int n, int& top) {
    uint32_t tp[P];
    planes8<P>(tword, tp);                                          // bit k of tp[b]: plane b of the word's k-th symbol
#pragma unroll
    for (int b = 0; b < P; b++) tp[b] = ~tp[b];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (!PRED || j0 + k + 1 <= n) {
            uint32_t nk[P];
#pragma unroll
            for (int b = 0; b < P; b++) nk[b] = (uint32_t)__builtin_amdgcn_sbfe((int)tp[b], k, 1);      // bit set -> 0, clear -> all ones: one v_bfe_i32
            top += 1;
            unsigned carry = 0, cyp = 1, cym = 0;                  // the row above the window steps +1
            uint32_t ph_last, mh_last;
            MYERS_COLUMN_C(Q, P, 1, pl, pv, mv, nk, carry, cyp, cym, ph_last, mh_last)
        }
    }
}

// per-lane bookkeeping of the staircase kernel
struct StairLane {
    int top;                    // D[row above the window][current column]
    int trow;                   // pattern row held by bit 0 of the window
    int kcap;                   // largest distance this attempt certifies
    int m, n, delta;
    int d;                      // the result, taken when the lane's last column is through (-1: row m was not inside the window)
    bool live;
    long long useful;           // word-columns the lane's pair needed
};

// Block kb (columns 32 kb + 1 .. 32 kb + 32) with Q live words, then the drop that prepares block kb + 1.  Returns the new Q.
// The window state is moved for ALL lanes alike (a lane-dependent move would make the compiler keep whole copies of the register arrays):
// a lane whose last column lies in this block takes its result here, before the drop.
template <int Q, int P, int QM>
__device__ __forceinline__ int stair_block(uint32_t (&pv)[QM], uint32_t (&mv)[QM], uint32_t (&pl)[P][QM], uint32_t (&tq)[4],
                                           const uint32_t (&pw_next)[4], StairLane& L, const int kb, const int nmin, const int nmax, const bool narrow,
                                           long long& issued) {
    const int n = L.n;
#pragma unroll 1
    for (int i = 0; i < 4; i++) {
        const int j0 = kb * 32 + i * 8;
        if (j0 >= nmax) break;
        const uint32_t tword = tq[0];
        tq[0] = tq[1]; tq[1] = tq[2]; tq[2] = tq[3];
        if (j0 + 8 <= nmin) stair_columns8<Q, P, false, QM>(pv, mv, pl, tword, j0, n, L.top);
        else stair_columns8<Q, P, true, QM>(pv, mv, pl, tword, j0, n, L.top);
    }
    {
        const int cols = nmax - kb * 32 >= 32 ? 32 : nmax - kb * 32;
        issued += (long long)cols * Q * 64;
        int mine = n - kb * 32; mine = mine > 32 ? 32 : (mine < 0 ? 0 : mine);
        L.useful += (long long)mine * Q;
    }
    const int j1 = 32 * (kb + 1);                                  // last column of the block
    const bool fin = L.live && n > 32 * kb && n <= j1;             // this lane's last column lies in this block
    if (__any(fin)) {
        // D[m][n] = D[row above the window][n] + vertical deltas of window bits 0 .. (m - first row of the window)
        const int bm = L.m - L.trow;
        int d = L.top;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int hi_bit = bm - 32 * q;
            const uint32_t mask = hi_bit >= 31 ? 0xffffffffu : (hi_bit < 0 ? 0u : ((2u << hi_bit) - 1u));
            d += __popc(pv[q] & mask) - __popc(mv[q] & mask);
        }
        if (fin) L.d = (bm >= 0 && bm < 32 * Q) ? d : -1;
    }
    if (j1 >= nmax) return Q;                                      // no lane has another block
    const bool active = L.live && j1 < n;
    const int t1 = L.top + __popc(pv[0]) - __popc(mv[0]);           // D[last row of word 0][j1]
    int t2 = t1;
    bool et = false, eb = false;
    if (narrow && Q > STAIR_QMIN) {
        t2 = t1 + __popc(pv[1]) - __popc(mv[1]);                    // D[last row of word 1][j1]
        const int oc_t = j1 - (L.trow + 63);                        // its diagonal
        const bool ok_t = !active || (oc_t >= L.delta && t2 + (oc_t - L.delta) > L.kcap);
        int sb = L.top;
#pragma unroll
        for (int q = 0; q < Q - 1; q++) sb += __popc(pv[q]) - __popc(mv[q]);          // D[last row of word Q-2][j1]
        const int bot = L.trow + 32 * Q;                            // first row below the window
        const int oc_b = j1 - (bot - 33);
        const bool ok_b = !active || bot > L.m || (oc_b <= L.delta && sb + (L.delta - oc_b) > L.kcap);
        et = __all(ok_t); eb = __all(ok_b);
        if (et && eb && Q - 2 < STAIR_QMIN) eb = false;
    }
    uint32_t acc[P];
    planes32<P>(pw_next[0], pw_next[1], pw_next[2], pw_next[3], acc);       // the word below the window (enters unless eb)
    if (et) {
        L.top = t2; L.trow += 64;
#pragma unroll
        for (int q = 0; q + 2 < Q; q++) {
            pv[q] = pv[q + 2]; mv[q] = mv[q + 2];
#pragma unroll
            for (int b = 0; b < P; b++) pl[b][q] = pl[b][q + 2];
        }
        if (!eb) {                                                  // a word that enters steps +1 per row
            pv[Q - 2] = 0xffffffffu; mv[Q - 2] = 0u;
#pragma unroll
            for (int b = 0; b < P; b++) pl[b][Q - 2] = acc[b];
        }
    } else {
        L.top = t1; L.trow += 32;
#pragma unroll
        for (int q = 0; q + 1 < Q; q++) {
            pv[q] = pv[q + 1]; mv[q] = mv[q + 1];
#pragma unroll
            for (int b = 0; b < P; b++) pl[b][q] = pl[b][q + 1];
        }
        if (!eb) {
            pv[Q - 1] = 0xffffffffu; mv[Q - 1] = 0u;
#pragma unroll
            for (int b = 0; b < P; b++) pl[b][Q - 1] = acc[b];
        }
    }
    return Q - (et ? 1 : 0) - (eb ? 1 : 0);
}

// wave-uniform state of a staircase run
struct StairRun {
    int Q, kb, n_blocks, nmin, nmax, txt_words, pat_words;
    bool narrow, live;
    long long issued;
};

// The number of live words only ever decreases: the run is a cascade of plain loops, one per Q, each entered when the window has that many
// words (a single loop around a switch over Q makes every register array a 14-way merge at its header).
template <int Q, int P, int QM>
__device__ __forceinline__ void stair_run(uint32_t (&pv)[QM], uint32_t (&mv)[QM], uint32_t (&pl)[P][QM], uint32_t (&tq)[4],
                                          uint32_t (&tw_next)[4], uint32_t (&pw_next)[4], StairLane& L, StairRun& R, const Packed& pat, const Packed& txt) {
    if (R.Q == Q && R.kb < R.n_blocks) {
#pragma unroll 1
        do {
            R.Q = stair_block<Q, P, QM>(pv, mv, pl, tq, pw_next, L, R.kb, R.nmin, R.nmax, R.narrow, R.issued);
            // text words of block kb + 1 (fetched a block ahead), then the fetches for block kb + 2 and for the pattern rows below the new window
            const int below = (L.trow + 32 * R.Q - 1) >> 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                tq[i] = tw_next[i];
                const int ti = 4 * (R.kb + 2) + i;
                tw_next[i] = (R.live && ti < R.txt_words) ? txt.word(ti) : 0u;
                const int pi = below + i;
                pw_next[i] = (R.live && pi >= 0 && pi < R.pat_words) ? pat.word(pi) : 0u;
            }
            R.kb++;
        } while (R.Q == Q && R.kb < R.n_blocks);
    }
    if constexpr (Q > STAIR_QMIN) stair_run<Q - 1, P, QM>(pv, mv, pl, tq, tw_next, pw_next, L, R, pat, txt);
}

template <int P, int QM>
__device__ __forceinline__ void d_edit_stair(const int Q0, const bool narrow, long long blk, long long count, const uint32_t* list, const uint32_t* scratch,
                                             PairDesc* desc, const long long* slot_of, int32_t* ed, unsigned long long* fail_cnt, uint32_t* fail_lists,
                                             long long fail_cap, unsigned long long* wc) {
    const long long t = blk * 256 + threadIdx.x;
    const bool live = t < count;
    uint32_t widx = 0;
    PairDesc pd; pd.m = 1; pd.n = 0; pd.pat = 0; pd.txt = 0; pd.ub = 0; pd.cls = 0;
    if (live) { widx = list[t]; pd = desc[widx]; }
    const int W0 = 32 * Q0;
    const int m = pd.m, n = live ? pd.n : 0;
    const int delta = n - m;
    int off = (W0 - 34 + delta) / 2;
    if (off < 7) off = 7;
    off -= ((off - 7) & 7);
    const int margin_lo = W0 - off - 33, margin_up = off + 1 - delta;
    const int margin = margin_lo < margin_up ? margin_lo : margin_up;
    const Packed pat{scratch + pd.pat, CLS_PAT_SH(pd.cls)}, txt{scratch + pd.txt, CLS_TXT_SH(pd.cls)};
    const int pat_words = (m + 7) >> 3, txt_words = (n + 7) >> 3;
    const int w0 = -((off + 1) >> 3);                   // packed pattern word that holds bit 0 of the window (row -off)
    auto pat_word = [&](int idx) -> uint32_t { return (live && idx >= 0 && idx < pat_words) ? pat.word(idx) : 0u; };
    // all QM words are set up, whatever Q0 is (the words beyond Q0 are never looked at): a set-up that depends on Q0 would make every
    // register array a union of its variants
    uint32_t pv[QM], mv[QM], pl[P][QM];
#pragma unroll
    for (int q = 0; q < QM; q++) {
        const int nvirt = off + 1 - 32 * q;             // bits of this word that are rows <= 0: vertical delta -1
        const uint32_t mlow = nvirt >= 32 ? 0xffffffffu : (nvirt <= 0 ? 0u : ((1u << nvirt) - 1u));
        mv[q] = mlow; pv[q] = ~mlow;
        uint32_t acc[P];
#pragma unroll
        for (int b = 0; b < P; b++) acc[b] = 0u;
        // (the value of a word beyond Q0 is never looked at - it only has to be SOME value in its register: a wave-uniform branch skips its four loads and ~60
        // instructions; the 3-word class spends 12 % less on its set-up)
        if (q < Q0) planes32<P>(pat_word(w0 + 4 * q), pat_word(w0 + 4 * q + 1), pat_word(w0 + 4 * q + 2), pat_word(w0 + 4 * q + 3), acc);
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = acc[b];
    }
    StairLane L;
    L.top = off + 1;                                     // D[row above the window][column]
    L.trow = -off; L.m = m; L.n = n; L.delta = delta; L.live = live; L.useful = 0; L.d = -1;
    // what the static window certifies: d <= delta + 2 margin + 1 (x = floor((d - delta) / 2) <= margin); an upper bound known beforehand
    // (trivial alignments, an earlier band's path) is never exceeded by the true distance, so it may stand in when it is smaller
    L.kcap = margin >= 0 ? delta + 2 * margin + 1 : -1;
    if (live && pd.ub < L.kcap) L.kcap = pd.ub;
    int nmax = n;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    // Columns are processed one packed text word (8 columns) per loop trip; the pairs of a wave are sorted by text length, so all but the
    // last few trips run without the per-column `j <= n` test (PRED = false).
    int nmin = live ? n : 0x7fffffff;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(nmin, o, 64); nmin = v < nmin ? v : nmin; }
    if (nmin == 0x7fffffff) nmin = 0;
    uint32_t tq[4], tw_next[4], pw_next[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        tq[i] = (live && i < txt_words) ? txt.word(i) : 0u;
        tw_next[i] = (live && 4 + i < txt_words) ? txt.word(4 + i) : 0u;
        pw_next[i] = pat_word(w0 + 4 * Q0 + i);                         // rows just below the window
    }
    StairRun R;
    R.Q = Q0; R.kb = 0; R.n_blocks = (nmax + 31) >> 5; R.nmin = nmin; R.nmax = nmax; R.narrow = narrow; R.issued = 0;
    R.live = live; R.txt_words = txt_words; R.pat_words = pat_words;
    stair_run<QM, P, QM>(pv, mv, pl, tq, tw_next, pw_next, L, R, pat, txt);
    const long long issued = R.issued;
    wc_account(wc, issued, L.useful);
    if (!live) return;
    const int d = L.d;
    if (margin >= 0 && d >= 0 && d <= L.kcap) ed[slot_of ? slot_of[widx] : (long long)widx] = d;
    else band_retry(pd, widx, m, n, d >= 0 ? d : pd.ub, desc, fail_cnt, fail_lists, fail_cap);
}

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v) {
    // lane l receives lane l-1's value, lane 0 receives 0 (gfx9 DPP control wave_shr:1)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true);       // bound_ctrl: no `old` value to set up - one v_mov_b32_dpp
}

// ---- 3b. whole pattern in one lane (m <= 32*Q), full matrix: plain multi-word Myers, 64 pairs per wave ----------
// The pattern is BOTTOM-aligned: row m is bit 31 of the last word, the 32*Q - m bits above row 1 are virtual rows <= 0
// (D[r][j] = j - r: vertical delta -1, horizontal +1 - a match there changes nothing).  The horizontal delta of row m is then
// simply the bit the shifts push out of the last word: no per-word test for "the word that holds row m".
// 8 symbols starting at core symbol s (any alignment); a chunk that lies entirely above the core (virtual rows) reads as 0, one that
// straddles its start picks up the symbols stored in front of the core - harmless there, see above
__device__ __forceinline__ uint32_t packed8_at(const Packed& pk, int s, bool live) {
    if (!live || s + 8 <= 0) return 0u;
    return fetch8(pk.w, s + (pk.sh4 >> 2));
}
// bit-plane words of the 32 rows whose first pattern symbol index is s
template <int P>
__device__ __forceinline__ void planes32_at(const Packed& pk, int s, bool live, uint32_t (&out)[P]) {
    planes32<P>(packed8_at(pk, s, live), packed8_at(pk, s + 8, live), packed8_at(pk, s + 16, live), packed8_at(pk, s + 24, live), out);
}

template <int Q, int P>
__device__ __forceinline__ void d_edit_lane(long long blk, long long count, const uint32_t* list, const uint32_t* scratch, const PairDesc* desc,
                                            const long long* slot_of, int32_t* ed, unsigned long long* wc) {
    const long long t = blk * 256 + threadIdx.x;
    const bool live = t < count;
    uint32_t widx = 0;
    PairDesc pd; pd.m = 1; pd.n = 0; pd.pat = 0; pd.txt = 0; pd.ub = 0; pd.cls = 0;
    if (live) { widx = list[t]; pd = desc[widx]; }
    const int m = pd.m, n = live ? pd.n : 0;
    const Packed pat{scratch + pd.pat, CLS_PAT_SH(pd.cls)}, txt{scratch + pd.txt, CLS_TXT_SH(pd.cls)};
    const int pad = 32 * Q - m;                          // virtual rows above row 1
    uint32_t pv[Q], mv[Q], pl[P][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int nvirt = pad - 32 * q;
        const uint32_t mlow = nvirt >= 32 ? 0xffffffffu : (nvirt <= 0 ? 0u : ((1u << nvirt) - 1u));
        mv[q] = mlow; pv[q] = ~mlow;
        uint32_t e[P];
        planes32_at<P>(pat, 32 * q - pad, live, e);
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = e[b];
    }
    int score = m;
    int nmax = n;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(nmax, o, 64); nmax = v > nmax ? v : nmax; }
    const int txt_words = (n + 7) >> 3;
    uint32_t tw_next = (live && txt_words > 0) ? txt.word(0) : 0u;
    for (int jb = 0; jb * 8 < nmax; jb++) {
        const uint32_t tw = tw_next;
        tw_next = (live && jb + 1 < txt_words) ? txt.word(jb + 1) : 0u;
        uint32_t tp[P];
        planes8<P>(tw, tp);                                             // bit k of tp[b]: plane b of the word's k-th symbol
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = jb * 8 + k + 1;
            if (j <= n) {
                uint32_t nk[P];
#pragma unroll
                for (int b = 0; b < P; b++) nk[b] = ((tp[b] >> k) & 1u) - 1u;
                unsigned carry = 0, cyp = 1, cym = 0;                   // the row above the column steps +1
                uint32_t ph_last, mh_last;
                MYERS_COLUMN_C(Q, P, 1, pl, pv, mv, nk, carry, cyp, cym, ph_last, mh_last)
                score += (int)cyp - (int)cym;                            // the bits pushed out of the last word: the horizontal delta of row m
            }
        }
    }
    wc_account(wc, (long long)nmax * Q * 64, (long long)n * Q);
    if (live) ed[slot_of ? slot_of[widx] : (long long)widx] = score;
}

// ---- 3d. full matrix in ROW BLOCKS: one lane per (pair, block of 32 Q rows), the blocks of a pair in successive launches (round 6, SVX_EDIT_BLOCKED=1) --------
// The multi-lane forms below sweep the text in lock step: a lane whose rows lie outside the band the pair's upper bound allows - the two corner triangles of a
// pair whose length gap exceeds every band - idles while its wave still pays for the column.  Here block r of a pair (rows r * 32 Q - pad + 1 .. of the
// bottom-aligned pattern, pad in block 0) is a task of its own: it walks only the columns jlo .. jhi its rows can meet on diagonals -x .. delta + x
// (x = floor((ub - delta) / 2) + 1: a path of cost <= ub leaves the corridor 0 .. delta by no more), 64 pairs per wave like d_edit_lane, no hand-off between
// lanes.  What it needs from the block above - the horizontal delta of that block's last row, two bits per column - comes through memory (written by the launch
// of rank r - 1; columns behind that block's range step +1), together with the value of its last row in front of this block's first column; the column in
// front of the block's range steps +1 per row.  Both assumptions are upper bounds of cells no path of cost <= ub visits (the cut-off argument of the staircase
// window), so the last block's last row ends at the distance.  Deltas stay in {-1, 0, +1}: the boundaries are.
struct BlkArgs { int rank; uint32_t* bnd_in; uint32_t* bnd_out; int* e_in; int* e_out; long long wmax; };
__device__ __forceinline__ long long blk_word_at(long long slot, long long wmax, int wi, int plane) { return (((slot >> 6) * wmax + wi) * 2 + plane) * 64 + (slot & 63); }

template <int Q, int P>
__device__ __forceinline__ void d_edit_blocked(const BlkArgs& A, long long slot_base, long long blk, long long count, const uint32_t* list, const uint32_t* scratch,
                                               const PairDesc* desc, const long long* slot_of, int32_t* ed, unsigned long long* wc) {
    constexpr int RL = 32 * Q;
    const long long t = blk * 256 + threadIdx.x;
    bool live = t < count;
    uint32_t widx = 0;
    PairDesc pd; pd.m = 1; pd.n = 1; pd.pat = 0; pd.txt = 0; pd.ub = 2; pd.cls = 0;
    if (live) { widx = list[t]; pd = desc[widx]; }
    const long long slot = slot_base + t;
    const int m = pd.m, n = pd.n, r = A.rank;
    const int B = (m + RL - 1) / RL;
    if (r >= B) live = false;
    const int pad = B * RL - m, delta = n - m;
    int ub = pd.ub; if (ub > m + n) ub = m + n; if (ub < delta) ub = delta;
    const int x = (ub - delta) / 2 + 1;
    const int ihi = (r + 1) * RL - pad, ilo = r * RL - pad + 1 > 1 ? r * RL - pad + 1 : 1;
    const int jlo = ilo - x > 1 ? ilo - x : 1;
    const long long jhi_l = (long long)ihi + delta + x;
    const int jhi = jhi_l > n ? n : (int)jhi_l;
    const long long jhp_l = (long long)(r * RL - pad) + delta + x;                       // last column of the block above
    const int jhi_prev = jhp_l > n ? n : (int)jhp_l;
    const int jlo_next = ihi + 1 - x > 1 ? ihi + 1 - x : 1;                              // first column of the block below
    const int len = live && jhi >= jlo ? jhi - jlo + 1 : 0;
    const Packed pat{scratch + pd.pat, CLS_PAT_SH(pd.cls)};
    const int ts = (CLS_TXT_SH(pd.cls) >> 2) + jlo - 1;                                 // text symbol of column jlo, counted from the record word pd.txt
    const Packed txt{scratch + pd.txt + (ts >> 3), (ts & 7) << 2};
    uint32_t pv[Q], mv[Q], pl[P][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int nvirt = r == 0 ? pad - 32 * q : 0;                                    // virtual rows above row 1 (block 0 only): vertical delta -1
        const uint32_t mlow = nvirt >= 32 ? 0xffffffffu : (nvirt <= 0 ? 0u : ((1u << nvirt) - 1u));
        mv[q] = mlow; pv[q] = ~mlow;
        uint32_t e[P];
        planes32_at<P>(pat, r * RL + 32 * q - pad, live, e);
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = e[b];
    }
    // value of the block's last row in front of its first column: column 0 (D[i][0] = i), or what the block above handed over + one per row
    int cur = jlo == 1 ? ihi : (live ? A.e_in[slot] : 0) + RL;
    int lenmax = len;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(lenmax, o, 64); lenmax = v > lenmax ? v : lenmax; }
    const int txt_words = (len + 7) >> 3;
    uint32_t tw_next = txt_words > 0 ? txt.word(0) : 0u;
    uint32_t in_p = 0u, in_m = 0u, out_p = 0u, out_m = 0u;
    const bool has_above = r > 0, has_below = r + 1 < B;
    for (int cb = 0; cb * 8 < lenmax; cb++) {
        const uint32_t tw = tw_next;
        tw_next = (cb + 1 < txt_words) ? txt.word(cb + 1) : 0u;
        uint32_t tp[P];
        planes8<P>(tw, tp);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int c = cb * 8 + k;
            if (c < len) {
                const int a = jlo - 1 + c, sh = a & 31;                                 // column j = a + 1; its bit in the boundary words
                unsigned cyp = 1, cym = 0;
                if (has_above) {
                    if (c == 0 || sh == 0) { in_p = A.bnd_in[blk_word_at(slot, A.wmax, a >> 5, 0)]; in_m = A.bnd_in[blk_word_at(slot, A.wmax, a >> 5, 1)]; }
                    if (a < jhi_prev) { cyp = (in_p >> sh) & 1u; cym = (in_m >> sh) & 1u; }
                }
                uint32_t nk[P];
#pragma unroll
                for (int b = 0; b < P; b++) nk[b] = ((tp[b] >> k) & 1u) - 1u;
                unsigned carry = 0;
                const uint32_t hin_m = cym;
                uint32_t ph_last, mh_last;
                MYERS_COLUMN_B(Q, P, pl, pv, mv, nk, hin_m, carry, cyp, cym, ph_last, mh_last)
                cur += (int)cyp - (int)cym;                                             // the bits pushed out of the last word: the horizontal delta of the block's last row
                if (has_below) {
                    out_p |= cyp << sh; out_m |= cym << sh;
                    if (sh == 31 || c == len - 1) {
                        A.bnd_out[blk_word_at(slot, A.wmax, a >> 5, 0)] = out_p; A.bnd_out[blk_word_at(slot, A.wmax, a >> 5, 1)] = out_m;
                        out_p = 0u; out_m = 0u;
                    }
                    if (a + 1 == jlo_next - 1) A.e_out[slot] = cur;
                }
            }
        }
    }
    wc_account(wc, (long long)lenmax * Q * 64, (long long)len * Q);
    if (live && !has_below) ed[slot_of ? slot_of[widx] : (long long)widx] = cur;
}

// ---- 3c. full matrix, G lanes per pair, 32 Q rows (Q = 16 or 12 words) per lane ------------------------------------------------
// The column is ONE wide bit-vector spread over the first L = ceil(m / (32 Q)) lanes of a group, bottom-aligned like above (row m is
// bit 31 of lane L-1's last word); lane g of a group works on text column t-g at step t and hands (symbol, adder carry,
// pushed-out plus/minus bits) to lane g+1 through a DPP wave shift.  64/G pairs share a wave, every lane carries 16 words of
// state, so the per-step overhead is amortised over 512 cells (the 1-block-per-lane systolic kernel pays it per 32 cells).
template <int G, int Q, int P>
__device__ __forceinline__ void d_edit_wide(long long blk, long long count, const uint32_t* list, const uint32_t* scratch, const PairDesc* desc,
                                            const long long* slot_of, int32_t* ed, unsigned long long* wc) {
    constexpr int RL = 32 * Q;                               // rows per lane
    const int lane = lane_id();
    const int gl = lane & (G - 1);
    const long long t = (blk * 256 + threadIdx.x) / G;
    const bool live = t < count;
    uint32_t widx = 0;
    PairDesc pd; pd.m = 1; pd.n = 0; pd.pat = 0; pd.txt = 0; pd.ub = 0; pd.cls = 0;
    if (live) { widx = list[t]; pd = desc[widx]; }
    const int m = pd.m, n = live ? pd.n : 0;
    const Packed pat{scratch + pd.pat, CLS_PAT_SH(pd.cls)}, txt{scratch + pd.txt, CLS_TXT_SH(pd.cls)};
    const int lanes_used = live ? (m + RL - 1) / RL : 0;
    const int pad = lanes_used * RL - m;                     // virtual rows above row 1 (all in lane 0)
    const int bit_base = gl * RL;                            // first bit of this lane in the wide vector
    uint32_t pv[Q], mv[Q], pl[P][Q];
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int nvirt = pad - (bit_base + 32 * q);
        const uint32_t mlow = nvirt >= 32 ? 0xffffffffu : (nvirt <= 0 ? 0u : ((1u << nvirt) - 1u));
        mv[q] = mlow; pv[q] = ~mlow;
        uint32_t e[P];
        planes32_at<P>(pat, bit_base + 32 * q - pad, live && gl < lanes_used, e);
#pragma unroll
        for (int b = 0; b < P; b++) pl[b][q] = e[b];
    }
    const bool is_last = gl == lanes_used - 1;               // this lane holds row m (in bit 31 of its last word)
    int score = m;                                           // every lane keeps one; the last lane's is the result
    int steps = live ? n + lanes_used - 1 : 0, smax = steps;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(smax, o, 64); smax = v > smax ? v : smax; }
    const int txt_words = (n + 7) >> 3;
    const bool mine = live && gl < lanes_used;               // this lane holds pattern rows
    const unsigned n_on = mine ? (unsigned)n : 0u;           // columns this lane works on (none: a lane beyond the pattern)
    // Round 6.  Every lane reads the text of its OWN column (column 8 jb + k - gl at sub-step k of trip jb): it keeps the plane bits of two consecutive packed
    // text words (16 columns) and shifts them so that sub-step k is bit k - the symbol masks no longer travel down the lanes (2 DPP moves + 2 selects per column
    // before).  What does travel down - the (plus, minus) bits leaving a lane's last row and the adder's carry - are carries, one bit per lane: they travel as
    // LANE MASKS in SGPR pairs, handed to the lane below by a scalar shift of the mask (3 DPP moves + 3 selects + 3 word-to-carry conversions before); the masks
    // feed the three carry chains of the column update directly (inverse ballot).
    const int wo = (gl + 7) >> 3, r8 = (8 - (gl & 7)) & 7;   // column of sub-step k = 8 (jb - wo) + r8 + k
    auto text_word = [&](int idx) -> uint32_t { return (mine && idx >= 0 && idx < txt_words) ? txt.word(idx) : 0u; };
    uint32_t tpa[P], tw_next = text_word(1 - wo);             // plane bits of word jb - wo; the packed word jb - wo + 1 (fetched a trip ahead)
    planes8<P>(text_word(-wo), tpa);
    const unsigned long long FIRST = __builtin_amdgcn_ballot_w64(gl == 0);      // lanes whose rows start at the top of a pattern: constant inputs
    unsigned long long m_ph = 0, m_mh = 0, m_cy = 0;         // bit L: what lane L pushed out at its last column (plus, minus, adder carry)
    uint32_t o_ph = 0u, o_mh = 0u;                           // a lane's last ph / mh words (bit 31 = the bits it hands down)
    unsigned o_cy = 0u;
    int col = -gl;                                           // text column of this lane at the current step
    for (int jb = 0; jb * 8 < smax; jb++) {
        uint32_t tpb[P], tp[P];
        planes8<P>(tw_next, tpb);
        tw_next = text_word(jb + 2 - wo);
#pragma unroll
        for (int b = 0; b < P; b++) { tp[b] = ~((tpa[b] | (tpb[b] << 8)) >> r8); tpa[b] = tpb[b]; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t nk[P];
#pragma unroll
            for (int b = 0; b < P; b++) nk[b] = (uint32_t)__builtin_amdgcn_sbfe((int)tp[b], k, 1);       // bit set -> 0, clear -> all ones
            // top row of a pattern: horizontal delta +1, no carry
            unsigned cyp = __builtin_amdgcn_inverse_ballot_w64((m_ph << 1) | FIRST), cym = __builtin_amdgcn_inverse_ballot_w64((m_mh << 1) & ~FIRST);
            unsigned carry = __builtin_amdgcn_inverse_ballot_w64((m_cy << 1) & ~FIRST);
            if ((unsigned)col < n_on) {
                MYERS_COLUMN_C(Q, P, 0, pl, pv, mv, nk, carry, cyp, cym, o_ph, o_mh)
                score += (int)(o_ph >> 31) - (int)(o_mh >> 31);
                o_cy = carry;
            }
            // (outside the branch: a mask assigned under divergent control flow would stop being a scalar)
            m_ph = __builtin_amdgcn_ballot_w64((int)o_ph < 0); m_mh = __builtin_amdgcn_ballot_w64((int)o_mh < 0); m_cy = __builtin_amdgcn_ballot_w64(o_cy != 0u);
            col += 1;
        }
    }
    wc_account(wc, (long long)smax * Q * 64, (live && gl == 0) ? (long long)n * ((m + 31) >> 5) : 0);
    if (live && is_last) ed[slot_of ? slot_of[widx] : (long long)widx] = score;
}

// ---- 4. full-matrix systolic kernel ----------------------------------------------------------------------------
__device__ __forceinline__ void myers_block(uint32_t eq, uint32_t& pv, uint32_t& mv, uint32_t& hp, uint32_t& hm, uint32_t topmask) {
    const uint32_t xv = eq | mv;
    eq |= hm;
    const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
    uint32_t ph = mv | ~(xh | pv);
    uint32_t mh = pv & xh;
    const uint32_t hpo = (ph & topmask) ? 1u : 0u, hmo = (mh & topmask) ? 1u : 0u;
    ph = (ph << 1) | hp;
    mh = (mh << 1) | hm;
    pv = mh | ~(xv | ph);
    mv = ph & xv;
    hp = hpo; hm = hmo;
}

// Exact distance between pat (length m >= 1, the shorter) and txt (length n >= m); whole wave cooperates.
// R = 32-row blocks per lane held in registers (m <= 64*32*R).
template <int R, int P>
__device__ int systolic_distance(const Packed& pat, int m, const Packed& txt, int n) {
    const int lane = lane_id();
    const int nb = (m + 31) >> 5;
    uint32_t pl[R][P], pv[R], mv[R], top[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int blk = lane * R + r;
        uint32_t acc[P];
#pragma unroll
        for (int b = 0; b < P; b++) acc[b] = 0u;
        if (blk < nb) {
            const int row0 = blk << 5;
            for (int i = 0; i < 32; i++) {
                const int row = row0 + i;
                if (row < m) {
                    const uint32_t c = sym<P>(pat.at(row));
#pragma unroll
                    for (int b = 0; b < P; b++) acc[b] |= ((c >> b) & 1u) << i;
                }
            }
        }
#pragma unroll
        for (int b = 0; b < P; b++) pl[r][b] = acc[b];
        pv[r] = 0xffffffffu; mv[r] = 0u;
        top[r] = (blk == nb - 1) ? (1u << ((m - 1) & 31)) : 0x80000000u;
    }
    const int lanes_used = (nb + R - 1) / R;
    const int last_lane = lanes_used - 1;
    const int last_r = (nb - 1) - last_lane * R;
    int score = m;
    const int steps = n + lanes_used - 1;
    uint32_t out = 0;
    uint32_t tc_next = (lane < n) ? txt.at(lane) : 0u;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const uint32_t tc = tc_next;
        const int nxt = t0 + 64 + lane;
        tc_next = (nxt < n) ? txt.at(nxt) : 0u;
#pragma unroll 4
        for (int j = 0; j < 64; j++) {
            const int t = t0 + j;
            if (t >= steps) break;
            uint32_t in = dpp_wave_shr1(out);
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)tc, j);
            if (lane == 0) in = (t < n) ? (sym<P>(c0) | 0x10u | 0x40u) : 0u;       // top row: horizontal delta +1
            if ((in & 0x40u) && lane < lanes_used) {
                const uint32_t c = in & 15u;
                uint32_t nk[P];
#pragma unroll
                for (int b = 0; b < P; b++) nk[b] = ((c >> b) & 1u) - 1u;
                uint32_t hp = (in >> 4) & 1u, hm = (in >> 5) & 1u;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (lane * R + r < nb) {
                        uint32_t eq = pl[r][0] ^ nk[0];
#pragma unroll
                        for (int b = 1; b < P; b++) eq &= pl[r][b] ^ nk[b];
                        myers_block(eq, pv[r], mv[r], hp, hm, top[r]);
                        if (lane == last_lane && r == last_r) score += (int)hp - (int)hm;
                    }
                }
                out = c | (hp << 4) | (hm << 5) | 0x40u;
            } else {
                out = 0;
            }
        }
    }
    return __shfl(score, last_lane, 64);
}

// cores longer than 64*32*8 rows: same systolic schedule, block state in a global scratch area (7 words per block)
__device__ int systolic_distance_big(const Packed& pat, int m, const Packed& txt, int n, uint32_t* state) {
    const int lane = lane_id();
    const int nb = (m + 31) >> 5;
    const int R = (nb + 63) / 64;
    for (int r = 0; r < R; r++) {
        const int blk = lane * R + r;
        if (blk >= nb) break;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, v = 0;
        const int row0 = blk << 5;
        for (int i = 0; i < 32; i++) {
            const int row = row0 + i;
            if (row < m) {
                const uint32_t c = pat.at(row);
                a0 |= (c & 1u) << i; a1 |= ((c >> 1) & 1u) << i; a2 |= ((c >> 2) & 1u) << i; a3 |= ((c >> 3) & 1u) << i;
                v |= 1u << i;
            }
        }
        uint32_t* s = state + (size_t)blk * 7;
        s[0] = a0; s[1] = a1; s[2] = a2; s[3] = a3; s[4] = v; s[5] = 0xffffffffu; s[6] = 0u;
    }
    const int lanes_used = (nb + R - 1) / R;
    const int last_lane = lanes_used - 1;
    int score = m;
    const int steps = n + lanes_used - 1;
    uint32_t out = 0;
    uint32_t tc_next = (lane < n) ? txt.at(lane) : 0u;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const uint32_t tc = tc_next;
        const int nxt = t0 + 64 + lane;
        tc_next = (nxt < n) ? txt.at(nxt) : 0u;
        for (int j = 0; j < 64; j++) {
            const int t = t0 + j;
            if (t >= steps) break;
            uint32_t in = dpp_wave_shr1(out);
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)tc, j);
            if (lane == 0) in = (t < n) ? (c0 | 0x10u | 0x40u) : 0u;
            if ((in & 0x40u) && lane < lanes_used) {
                const uint32_t c = in & 15u;
                const uint32_t n0 = (c & 1u) - 1u, n1 = ((c >> 1) & 1u) - 1u, n2 = ((c >> 2) & 1u) - 1u, n3 = ((c >> 3) & 1u) - 1u;
                uint32_t hp = (in >> 4) & 1u, hm = (in >> 5) & 1u;
                for (int r = 0; r < R; r++) {
                    const int blk = lane * R + r;
                    if (blk >= nb) break;
                    uint32_t* s = state + (size_t)blk * 7;
                    const uint32_t eq = (s[0] ^ n0) & (s[1] ^ n1) & (s[2] ^ n2) & (s[3] ^ n3) & s[4];
                    uint32_t pv = s[5], mv = s[6];
                    const uint32_t top = (blk == nb - 1) ? (1u << ((m - 1) & 31)) : 0x80000000u;
                    myers_block(eq, pv, mv, hp, hm, top);
                    s[5] = pv; s[6] = mv;
                    if (blk == nb - 1) score += (int)hp - (int)hm;
                }
                out = c | (hp << 4) | (hm << 5) | 0x40u;
            } else {
                out = 0;
            }
        }
    }
    return __shfl(score, last_lane, 64);
}

// one wave per pair of the FULL class; pairs with more than 16384 rows are deferred to k_edit_full_big
template <int P>
__device__ __forceinline__ void d_edit_full(long long blk, long long count, const uint32_t* list, const uint32_t* scratch, const PairDesc* desc,
                                                   const long long* slot_of, int32_t* ed, unsigned long long* n_big, uint32_t* big_list, unsigned long long* wc) {
    const long long t = blk * 4 + (threadIdx.x >> 6);
    if (t >= count) return;
    const uint32_t widx = list[t];
    const PairDesc pd = desc[widx];
    Packed PP{scratch + pd.pat, CLS_PAT_SH(pd.cls)}, T{scratch + pd.txt, CLS_TXT_SH(pd.cls)};
    const int nb = (pd.m + 31) >> 5;
    int d;
    if (nb <= 64) d = systolic_distance<1, P>(PP, pd.m, T, pd.n);
    else if (nb <= 128) d = systolic_distance<2, P>(PP, pd.m, T, pd.n);
    else if (nb <= 256) d = systolic_distance<4, P>(PP, pd.m, T, pd.n);
    else if (nb <= 512) d = systolic_distance<8, P>(PP, pd.m, T, pd.n);
    else { if (lane_id() == 0) { const unsigned long long i = atomicAdd(n_big, 1ull); big_list[i] = widx; } return; }
    {
        const int R = nb <= 64 ? 1 : nb <= 128 ? 2 : nb <= 256 ? 4 : 8;
        wc_account(wc, (long long)(pd.n + (nb + R - 1) / R - 1) * R * 64, lane_id() == 0 ? (long long)pd.n * nb : 0);
    }
    if (lane_id() == 0) ed[slot_of ? slot_of[widx] : (long long)widx] = d;
}

// ---- 5. one launch per round and kind ---------------------------------------------------------------------------------
// The classes of a round are independent.  Launching them as separate kernels leaves the chip under-filled whenever a class is
// small or its last waves drag on (and the runtime multiplexes streams onto a handful of hardware queues), so a round is TWO
// launches: every band class in one grid, every full-matrix class in another.  Segments are laid out costliest first; a block
// looks up its segment (uniform, scalar) and runs that class's routine.
#define SEG_MAX 22
// A full-matrix pair is one serial chain of n steps, each as long as the words a lane holds.  When a round has only a few hundred long
// pairs (HiFi-like data: related pairs need tiny bands, what is left are a few unrelated long ones) the launch lasts as long as its
// slowest wave while most of the chip idles; the host then launches the classes beyond 2048 rows in their low-latency form - 64 lanes
// with 2-4 words each instead of 8-16 lanes with 12-16 words: ~3x shorter chains for ~1.4x the instructions.
#define KIND_LL 32
// the band pairs of a small retry round (run_edit_pipeline): one wave per pair, the form chosen by the pair's own rows - the 64-lane low-latency forms up to
// 8192 rows (a 631 x 725 pair: 0.06 ms against 0.21 as a systolic matrix and 0.31 as a 16-word band in one lane), the systolic kernel beyond
#define KIND_RETRY 40
struct FusedTab {
    int n;
    int narrow;                            // band launches: the staircase windows narrow as they run (d_edit_stair)
    int kind[SEG_MAX];                     // class id, or KIND_LL + words for a low-latency form
    unsigned first_block[SEG_MAX + 1];
    long long lo[SEG_MAX], cn[SEG_MAX];    // range of the class in the sorted list
};

template <int P, int QM>
__global__ __launch_bounds__(256) void k_edit_bands(FusedTab tab, const uint32_t* list, const uint32_t* scratch, PairDesc* desc,
                                                    const long long* slot_of, int32_t* ed, unsigned long long* fail_cnt, uint32_t* fail_lists,
                                                    long long fail_cap, unsigned long long* wc) {
    int s = 0;
    while (s + 1 < tab.n && blockIdx.x >= tab.first_block[s + 1]) s++;
    const long long blk = (long long)(blockIdx.x - tab.first_block[s]);
    const uint32_t* l = list + tab.lo[s];
    switch (tab.kind[s]) {
        case 0: d_edit_band<1, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, fail_cnt, fail_lists, fail_cap, wc); break;
                                                default: d_edit_stair<P, QM>(band_words(tab.kind[s]), tab.narrow != 0, blk, tab.cn[s], l, scratch, desc, slot_of, ed, fail_cnt, fail_lists, fail_cap, wc); break;
    }
}

template <int P>
__global__ __launch_bounds__(256) void k_edit_fulls(FusedTab tab, const uint32_t* list, const uint32_t* scratch, const PairDesc* desc,
                                                    const long long* slot_of, int32_t* ed, unsigned long long* n_big, uint32_t* big_list, unsigned long long* wc) {
    int s = 0;
    while (s + 1 < tab.n && blockIdx.x >= tab.first_block[s + 1]) s++;
    const long long blk = (long long)(blockIdx.x - tab.first_block[s]);
    const uint32_t* l = list + tab.lo[s];
    int kind = tab.kind[s];
    if (kind == KIND_RETRY) {                               // (uniform per wave: every wave of the segment has one pair)
        const long long t = blk * 4 + (long long)(threadIdx.x >> 6);
        const int m = __builtin_amdgcn_readfirstlane(t < tab.cn[s] ? desc[l[t]].m : 0);
        kind = m <= 4096 ? KIND_LL + 2 : (m <= 6144 ? KIND_LL + 3 : (m <= 8192 ? KIND_LL + 4 : (int)CLS_FULL));
    }
    switch (kind) {
        case CLS_FULL: d_edit_full<P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, n_big, big_list, wc); break;
        case CLS_LANE0: d_edit_lane<1, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_LANE0 + 1: d_edit_lane<2, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_LANE0 + 2: d_edit_lane<4, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_LANE0 + 3: d_edit_lane<8, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_LANE0 + 4: d_edit_lane<16, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE0: d_edit_wide<2, 16, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE0 + 1: d_edit_wide<4, 16, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE0 + 2: d_edit_wide<8, 16, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE0 + 3: d_edit_wide<16, 16, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE12: d_edit_wide<2, 12, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE12 + 1: d_edit_wide<4, 12, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE12 + 2: d_edit_wide<8, 12, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE12 + 3: d_edit_wide<16, 12, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE14: d_edit_wide<2, 14, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE14 + 1: d_edit_wide<4, 14, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE14 + 2: d_edit_wide<8, 14, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE14 + 3: d_edit_wide<16, 14, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE10: d_edit_wide<2, 10, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE10 + 1: d_edit_wide<4, 10, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE10 + 2: d_edit_wide<8, 10, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case CLS_WIDE10 + 3: d_edit_wide<16, 10, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        // low-latency forms of the long classes (KIND_LL + words per lane): one pair per wave, 64 lanes x 2 / 3 / 4 words
        case KIND_LL + 2: d_edit_wide<64, 2, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case KIND_LL + 3: d_edit_wide<64, 3, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        default: d_edit_wide<64, 4, P>(blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
    }
}

// the row-block launches of the classes below (SVX_EDIT_BLOCKED): one launch per block rank, segments as in k_edit_fulls; slot_base[s] = position of the
// segment's first pair among the blocked pairs of the round (the index of its boundary words)
struct BlkTab { long long slot_base[SEG_MAX]; };
template <int P>
__global__ __launch_bounds__(256) void k_edit_blocked(FusedTab tab, BlkTab bt, BlkArgs A, const uint32_t* list, const uint32_t* scratch, const PairDesc* desc,
                                                      const long long* slot_of, int32_t* ed, unsigned long long* wc) {
    int s = 0;
    while (s + 1 < tab.n && blockIdx.x >= tab.first_block[s + 1]) s++;
    const long long blk = (long long)(blockIdx.x - tab.first_block[s]);
    const uint32_t* l = list + tab.lo[s];
    switch (tab.kind[s]) {
        case 10: d_edit_blocked<10, P>(A, bt.slot_base[s], blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case 12: d_edit_blocked<12, P>(A, bt.slot_base[s], blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        case 14: d_edit_blocked<14, P>(A, bt.slot_base[s], blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
        default: d_edit_blocked<16, P>(A, bt.slot_base[s], blk, tab.cn[s], l, scratch, desc, slot_of, ed, wc); break;
    }
}


__global__ __launch_bounds__(64) void k_edit_full_big(long long count, const uint32_t* big_list, const long long* state_off, const uint32_t* scratch,
                                                      const PairDesc* desc, const long long* slot_of, int32_t* ed, uint32_t* state) {
    const long long q = blockIdx.x;
    if (q >= count) return;
    const uint32_t widx = big_list[q];
    const PairDesc pd = desc[widx];
    Packed P{scratch + pd.pat, CLS_PAT_SH(pd.cls)}, T{scratch + pd.txt, CLS_TXT_SH(pd.cls)};
    const int d = systolic_distance_big(P, pd.m, T, pd.n, state + state_off[q]);
    if (lane_id() == 0) ed[slot_of ? slot_of[widx] : (long long)widx] = d;
}

__global__ void k_slots(long long n_work, PairSource src, long long* slot_of) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_work) slot_of[w] = src.slot(w);
}

// class boundaries in the sorted key array: first index whose sort class >= c, for c = 0..N_SORT_CLASSES
// threads 128 .. 191 (row-block route): split[c] = first index of class c whose cost order is at most `limit` (the pairs of a class are sorted by descending cost)
__global__ void k_class_bounds(const uint64_t* keys, long long n, long long* bounds, long long* split, long long limit) {
    const int c = threadIdx.x;
    if (c >= 128 && c < 128 + N_SORT_CLASSES && split) {
        const unsigned long long top = (1ull << SORT_COST_BITS) - 1, want = sort_key_of((unsigned long long)(c - 128), top - (unsigned long long)(limit < 0 ? 0 : ((unsigned long long)limit > top ? top : limit)));
        long long lo = 0, hi = n;
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
        split[c - 128] = lo;
        return;
    }
    if (c > N_SORT_CLASSES) return;
    long long lo = 0, hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)(keys[mid] >> SORT_COST_BITS) < c) lo = mid + 1; else hi = mid; }
    bounds[c] = lo;
}

// ---- 6. divergence histogram ------------------------------------------------------------------------------------------
// (d - (n-m)) / m of every pair with a core of at least 128 symbols, 256 bins, weighted by the core length: what the band
// speculation of the NEXT call is calibrated with (svx_ctx::edit_guess)
__global__ __launch_bounds__(256) void k_edit_hist(long long n_work, const PairDesc* desc, const long long* slot_of, const int32_t* ed, unsigned long long* hist) {
    __shared__ unsigned long long h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < n_work; w += (long long)gridDim.x * 256) {
        const PairDesc pd = desc[w];
        if (pd.m < 128) continue;
        const long long x = (long long)ed[slot_of[w]] - (pd.n - pd.m) - 2 * CLS_SHIFT(pd.cls);
        long long b = x * 256 / pd.m;
        if (b < 0) b = 0;
        if (b > 255) b = 255;
        atomicAdd(&h[b], (unsigned long long)pd.m);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(hist + threadIdx.x, h[threadIdx.x]);
}

// ---- host orchestration -----------------------------------------------------------------------------------------
// SVX_EDIT_PROFILE=1: per round and class, the pair count and the 32-bit word-columns the class kernel executes
// (useful = sum over pairs, issued = what the lock-stepped waves pay: 64 x the longest text of each wave); stderr, one line per class
static void profile_round(svx_ctx* c, int round, const long long* seg_lo, const long long* seg_cn, const uint32_t* list_dev, const PairDesc* desc_dev, long long n_desc) {
    hipStream_t st = c->stream;
    std::vector<PairDesc> desc((size_t)n_desc);
    if (svx_d2h(desc.data(), desc_dev, (size_t)n_desc * sizeof(PairDesc), st) != SVX_OK) return;
    for (int sc = 0; sc < N_SORT_CLASSES; sc++) {
        const int cls = sc & (GENERIC_BASE - 1), generic = sc / GENERIC_BASE;
        const long long cn = seg_cn[sc];
        if (cls >= N_CLASSES || cn <= 0) continue;
        std::vector<uint32_t> list((size_t)cn);
        if (svx_d2h(list.data(), list_dev + seg_lo[sc], (size_t)cn * 4, st) != SVX_OK) return;
        int words = 0, per_wave = 64;                       // 32-bit words of column state per pair; pairs per wave
        if (cls < NBAND) words = band_words(cls);
        else if (cls >= CLS_LANE0 && cls < CLS_WIDE0) words = 1 << (cls - CLS_LANE0);
        else if (cls >= CLS_WIDE10) { words = 10 * (2 << (cls - CLS_WIDE10)); per_wave = 64 / (2 << (cls - CLS_WIDE10)); }
        else if (cls >= CLS_WIDE14) { words = 14 * (2 << (cls - CLS_WIDE14)); per_wave = 64 / (2 << (cls - CLS_WIDE14)); }
        else if (cls >= CLS_WIDE12) { words = 12 * (2 << (cls - CLS_WIDE12)); per_wave = 64 / (2 << (cls - CLS_WIDE12)); }
        else if (cls >= CLS_WIDE0) { words = 16 * (2 << (cls - CLS_WIDE0)); per_wave = 64 / (2 << (cls - CLS_WIDE0)); }
        double useful = 0, issued = 0, sum_m = 0, sum_n = 0;
        for (long long i = 0; i < cn; i += per_wave) {
            long long nmax = 0;
            for (long long k = i; k < cn && k < i + per_wave; k++) {
                const PairDesc& pd = desc[list[(size_t)k]];
                const int w = cls == CLS_FULL ? (pd.m + 31) / 32 : words;
                useful += (double)pd.n * w; sum_m += pd.m; sum_n += pd.n;
                if (cls == CLS_FULL) issued += (double)pd.n * w; else if (pd.n > nmax) nmax = pd.n;
            }
            if (cls != CLS_FULL) issued += (double)nmax * words * per_wave;
        }
        fprintf(stderr, "{\"edit_profile\": {\"round\": %d, \"cls\": %d, \"generic\": %d, \"pairs\": %lld, \"mean_m\": %.1f, \"mean_n\": %.1f, \"word_cols_useful\": %.4g, \"word_cols_issued\": %.4g}}\n",
                round, cls, generic, cn, sum_m / cn, sum_n / cn, useful, issued);
    }
}

#define MAX_ROUNDS 8

// the speculation fraction g with the least expected cost over a (core-length weighted) divergence histogram of 256 bins:
// the first band costs ~g per cell column; a pair beyond g retries at 2g, 4g, ... (a band wider than half the core is a full matrix, ~0.75)
static float guess_from_histogram(const unsigned long long* h, float fallback) {
    double total = 0, best_cost = 1e300;
    double cum[257];
    cum[0] = 0;
    for (int b = 0; b < 256; b++) { total += (double)h[b]; cum[b + 1] = total; }
    if (total <= 0) return fallback;
    auto beyond = [&](double g) { int b = (int)(g * 256.0); if (b > 256) b = 256; return 1.0 - cum[b] / total; };
    int best = 31;
    for (int b = 3; b < 128; b++) {
        const double g = (b + 1) / 256.0;
        double cost = g;
        for (double w = g; w < 1.0; w *= 2) cost += (2 * w <= 0.5 ? 2 * w : 0.75) * beyond(w);
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return (float)((best + 2) / 256.0);
}

static int run_edit_pipeline(svx_ctx* c, long long n_work, const PairSource& src_in, int32_t* ed_dev, unsigned long long* cells_dev) {
    if (n_work <= 0) return SVX_OK;
    if (n_work >= (1ll << 32)) return svx_fail(SVX_E_ARG, "more than 2^32 edit-distance pairs in one call", __FILE__, __LINE__, hipSuccess);
    hipStream_t st = c->stream;
    const int T = 256;
    PairSource src = src_in;
    src.n_pairs = n_work;
    const size_t WC_OFF = 128 + MAX_ROUNDS * N_SORT_CLASSES, WC_PER_LAUNCH = 2 * WC_SHARDS, EARLY_OFF = WC_OFF + MAX_ROUNDS * 2 * WC_PER_LAUNCH, CNT_WORDS = EARLY_OFF + 2 * N_SORT_CLASSES;
    SVXCHK(c->e_fail.reserve(CNT_WORDS * 8));
    unsigned long long* cnt = c->e_fail.as<unsigned long long>();        // [1] big pairs, [8..40] class bounds of round 0 (first [16..31] the span shards), [128 + 64 r ..] retry counters written by round r, [WC_OFF ..] word-column counters per (round, kind)
    HIPCHK(hipMemsetAsync(cnt, 0, CNT_WORDS * 8, st));
    // cnt[2]: the speculation fraction k_edit_classify reads (float bits) - the fallback here, overwritten by k_edit_guess when the call is sampled
    HIPCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cnt + 2), (int)__builtin_bit_cast(uint32_t, c->edit_guess), 1, st));
    // 1. packed store: one record per string / signature
    src.radius = 0;
    const bool prepacked = !src.plain && c->prepack_state == 2 && c->prepack_n == src.in.n;
    c->prepack_state = 0;
    if (prepacked) {
        src.radius = c->prepack_radius;
        HIPCHK(hipStreamWaitEvent(st, c->ev[18], 0));
    } else if (!src.plain) {
        k_pair_span<<<(unsigned)(c->n_cu * 4), T, 0, st>>>(n_work, src, cnt);
        unsigned long long shard[16], span = 0;
        SVXCHK(svx_mail_read(c, st, cnt + 16, 16, shard));
        for (unsigned long long v : shard) span = v > span ? v : span;
        src.radius = (long long)span + 100;
    }
    const long long n_rec = src.plain ? 2 * n_work : src.in.n;
    int64_t total_words = 0;
    if (prepacked) {
        src.rec = c->e_rec.as<HapRec>();
        total_words = c->pinned[0];
    } else {
        SVXCHK(c->e_words.reserve((size_t)(n_rec + 1) * 8));
        SVXCHK(c->e_off.reserve((size_t)(n_rec + 1) * 8));
        SVXCHK(c->e_rec.reserve((size_t)n_rec * sizeof(HapRec) + 64));
        src.rec = c->e_rec.as<HapRec>();
        k_hap_words<<<(unsigned)((n_rec + 1 + T - 1) / T), T, 0, st>>>(n_rec, src, c->e_words.as<int64_t>());
        SVXCHK(svx_exclusive_scan_i64(c, c->e_words.as<int64_t>(), c->e_off.as<int64_t>(), n_rec + 1));
        SVXCHK(svx_mail_read(c, st, c->e_off.as<int64_t>() + n_rec, 1, &total_words));
        SVXCHK(c->e_scratch.reserve((size_t)(total_words + 64) * 4));
        HIPCHK(hipMemsetAsync(c->e_scratch.p, 0, 4, st));                    // the leading pad word
        k_hap_pack<<<(unsigned)((n_rec + 3) / 4), 256, 0, st>>>(n_rec, src, c->e_off.as<int64_t>(), c->e_scratch.as<uint32_t>(), c->e_rec.as<HapRec>());
        HIPCHK(hipGetLastError());
    }
    SVXCHK(c->e_desc.reserve((size_t)n_work * sizeof(PairDesc)));
    SVXCHK(c->e_key.reserve((size_t)n_work * 8 * 2)); SVXCHK(c->e_val.reserve((size_t)n_work * 4 * 2));
    SVXCHK(c->e_slot.reserve((size_t)n_work * 8));
    SVXCHK(c->e_big_list.reserve((size_t)n_work * 4 + 64));
    uint64_t* key_a = c->e_key.as<uint64_t>(); uint64_t* key_b = key_a + n_work;
    uint32_t* val_a = c->e_val.as<uint32_t>(); uint32_t* val_b = val_a + n_work;
    long long* slot_of = c->e_slot.as<long long>();
    PairDesc* desc = c->e_desc.as<PairDesc>();
    uint32_t* scratch = c->e_scratch.as<uint32_t>();
    c->stats.n_hap_bytes += total_words * 4;
    // 2. trim + classify
    int shift_bounds = 1;                                   // SVX_EDIT_SHIFT_BOUNDS=0: upper bounds from the left-justified alignment only (A/B switch)
    if (const char* e = getenv("SVX_EDIT_SHIFT_BOUNDS")) shift_bounds = atoi(e) == 0 ? 0 : 1;
    k_slots<<<(unsigned)((n_work + T - 1) / T), T, 0, st>>>(n_work, src, slot_of);
    k_edit_prep<PREP_LANES><<<(unsigned)((n_work + 4 * (64 / PREP_LANES) - 1) / (4 * (64 / PREP_LANES))), 256, 0, st>>>(n_work, src, scratch, desc, key_a, val_a, ed_dev, cells_dev, shift_bounds);
    HIPCHK(hipGetLastError());
    const bool profile = getenv("SVX_EDIT_PROFILE") != nullptr, serial = getenv("SVX_EDIT_SERIAL") != nullptr;
    std::vector<PairDesc> first_desc;                       // SVX_EDIT_PROFILE: the descriptors as round 0 saw them (first class of every pair)
    // 2a. band speculation of THIS call from a strided sample of its own pairs (k_edit_pilot); SVX_EDIT_GUESS pins it instead
    float guess = c->edit_guess;
    if (!c->edit_guess_pinned && !c->edit_force_full && n_work >= 4096) {
        SVXCHK(c->e_hist.reserve(256 * 8));
        HIPCHK(hipMemsetAsync(c->e_hist.p, 0, 256 * 8, st));
        const long long stride = (n_work + PILOT_MAX - 1) / PILOT_MAX, n_samp = (n_work + stride - 1) / stride;
        k_edit_pilot<<<(unsigned)((n_samp + 255) / 256), 256, 0, st>>>(n_work, stride, desc, scratch, c->e_hist.as<unsigned long long>());
        k_edit_guess<<<1, 128, 0, st>>>(c->e_hist.as<unsigned long long>(), cnt + 2);        // the host learns the value with the last post of the call (k_edit_tail)
        HIPCHK(hipGetLastError());
        if (profile) {
            unsigned long long gw = 0;
            SVXCHK(svx_mail_read(c, st, cnt + 2, 1, &gw));
            guess = __builtin_bit_cast(float, (uint32_t)gw);
            fprintf(stderr, "{\"edit_guess_pilot\": %.4f, \"sampled_pairs\": %lld}\n", guess, n_samp);
        }
    }
    int narrow_windows = 1;                                 // SVX_EDIT_NARROW=0: static staircase windows (A/B switch; results are the same either way)
    if (const char* e = getenv("SVX_EDIT_NARROW")) narrow_windows = atoi(e) == 0 ? 0 : 1;
    long long few_pairs = 2048;                             // SVX_EDIT_FEW_PAIRS: calls with at most this many pairs take the low-latency route (0 = never)
    if (const char* e = getenv("SVX_EDIT_FEW_PAIRS")) few_pairs = atoll(e);
    // SVX_EDIT_BLOCKED=1 (round 6, opt-in): the round-0 pairs of the 2- and 4-lane full-matrix families run in row blocks that skip the corners outside the band of
    // the pair's upper bound (d_edit_blocked); SVX_EDIT_BLOCKED_K: up to which family (0: 2 lanes, 1: 4 lanes [default], 2: 8 lanes)
    int blocked_kmax = -1;
    if (const char* e = getenv("SVX_EDIT_BLOCKED")) if (e[0] == '1' && !c->edit_force_full && n_work > few_pairs) { blocked_kmax = 1; if (const char* k = getenv("SVX_EDIT_BLOCKED_K")) blocked_kmax = atoi(k) < 0 ? 0 : (atoi(k) > 2 ? 2 : atoi(k)); }
    k_edit_classify<<<(unsigned)((n_work + T - 1) / T), T, 0, st>>>(n_work, desc, key_a, val_a, c->edit_force_full ? 1 : 0, cnt + 2, (n_work <= few_pairs ? 1 : 0) | ((blocked_kmax + 1) << 1));
    HIPCHK(hipGetLastError());
    // 3. group by class (and by descending text length inside a class, so that the 64 pairs of a wave finish together)
    SVXCHK(svx_sort_pairs_u64(c, key_a, key_b, val_a, val_b, n_work, 0, SORT_COST_BITS + 7));      // 17 bits of cost order + 7 bits of class: three passes
    // the row-block route takes the pairs of its classes that walk at most this many columns per block (the blocks of a pair are serial launches: the longest
    // texts would make the chain of launches the tail of the window) - the others stay in the multi-lane forms (SVX_EDIT_BLOCKED_WALK)
    long long blocked_walk = 2500;
    if (const char* e = getenv("SVX_EDIT_BLOCKED_WALK")) blocked_walk = atoll(e);
    SVXCHK(c->e_hist.reserve(256 * 8));
    k_class_bounds<<<1, 256, 0, st>>>(key_b, n_work, reinterpret_cast<long long*>(cnt + 8), blocked_kmax >= 0 ? c->e_hist.as<long long>() : nullptr, blocked_walk);
    long long bounds[N_SORT_CLASSES + 1], blocked_split[N_SORT_CLASSES];
    unsigned long long blocked_nmax = 0;
    if (blocked_kmax >= 0) SVXCHK(svx_mail_read3(c, st, cnt + 8, N_SORT_CLASSES + 1, bounds, cnt + 3, 1, &blocked_nmax, c->e_hist.p, N_SORT_CLASSES, blocked_split));
    else SVXCHK(svx_mail_read(c, st, cnt + 8, N_SORT_CLASSES + 1, bounds));
    // A retry round with only a few band pairs is bound by the LATENCY of its slowest pair: a band retry walks its columns in one lane with 6-16 state words
    // (configs[1], round 1: 196 pairs, one wave of 16-word windows = 0.31 ms at the very end of the window), a full matrix spread over the 64 lanes of a wave
    // (KIND_RETRY) finishes the same pair in a fifth of that and never fails, so the round is also the last.  Up to this many band pairs of a retry round go there
    // (SVX_EDIT_RETRY_FULL, 0 = never; the distances are exact either way).
    long long retry_full = 4096;
    if (const char* e = getenv("SVX_EDIT_RETRY_FULL")) retry_full = atoll(e);

    // 4. rounds.  Full-matrix classes (never fail, most of the work and the longest serial chains) run on high-priority streams, band classes (may fail)
    // on low-priority ones (api.hip, SVX_EDIT_PRIO), each kind as ONE fused launch per round.  A failing pair is appended to the retry list of its next
    // class; a round only waits for its band launch, reads 64 counters and launches the next round straight from those lists - no sort, no small kernels
    // that would queue behind the long-running waves - so the retry rounds overlap the full-matrix work of the earlier ones.
    // (Measured and taken out again, profiles/r04_edit_early_fulls_ab.txt: the pairs whose length gap alone exceeds every band - known before
    // k_edit_prep has read a symbol, most of the full-matrix work - launched while bounds, pilot and sort of the other pairs are still computed.  The
    // full-matrix waves leave k_edit_prep one or two waves per SIMD: it takes 6.4 ms instead of 1.6, the band launches start 5 ms later and the window
    // ends where it ended before.)
    hipStream_t band_st[2] = {c->aux[0], c->aux[1]};     // A/C/G/T-only pairs, generic pairs
    hipStream_t full_st[2] = {c->aux[2], c->aux[3]};     // even / odd rounds
    long long seg_lo[N_SORT_CLASSES], seg_cn[N_SORT_CLASSES];
    long long pending = 0;
    for (int sc = 0; sc < N_SORT_CLASSES; sc++) { seg_lo[sc] = bounds[sc]; seg_cn[sc] = bounds[sc + 1] - bounds[sc]; pending += seg_cn[sc]; }
    const uint32_t* list = val_b;
    bool blocked_used = false;
    for (int round = 0; pending > 0; round++) {
        if (round >= MAX_ROUNDS) return svx_fail(SVX_E_STATE, "edit-distance retry loop did not converge", __FILE__, __LINE__, hipSuccess);
        if (profile) profile_round(c, round, seg_lo, seg_cn, list, desc, n_work);
        if (profile && round == 0) {
            first_desc.resize((size_t)n_work);
            SVXCHK(svx_d2h(first_desc.data(), desc, (size_t)n_work * sizeof(PairDesc), st));
        }
        // retry lists this round's band launch appends to: one list of `pending` slots per sort class.  Three buffers rotate; the one reused now was
        // last read by round-2's launches
        DevBuf& fb = c->e_retry[round % 3];
        if (round >= 2) HIPCHK(hipStreamSynchronize(full_st[round & 1]));
        // (two sets of lists: the second one and the counters at EARLY_OFF belong to the FIRST PART of round 0 alone.  Its failures are launched as soon as that
        // part's kernels have ended - their list entries are then visible - while the second part is still appending to the round's own lists: a counter of those
        // says how many entries have been CLAIMED, not how many have been written, and a store of a running kernel need not have left its XCD's L2.  Until round 6
        // both parts shared lists and the early launch took "what the counters say now": entries of the second part in flight among them.)
        SVXCHK(fb.reserve((size_t)3 * N_SORT_CLASSES * (size_t)pending * 4 + 64));
        uint32_t* const fb_early[2] = {fb.as<uint32_t>() + (size_t)N_SORT_CLASSES * (size_t)pending, fb.as<uint32_t>() + (size_t)2 * N_SORT_CLASSES * (size_t)pending};
        unsigned long long* const fail_cnt_early[2] = {cnt + EARLY_OFF, cnt + EARLY_OFF + N_SORT_CLASSES};
        hipStream_t const early_st[2] = {c->aux[5], c->aux[7]};          // the early parts of the two alphabets, each a chain of its own (both high priority)
        struct EarlyPost { unsigned long long ticket; int generic, set; };
        std::vector<EarlyPost> early_posts;
        unsigned long long* fail_cnt = cnt + 128 + (size_t)round * N_SORT_CLASSES;
        unsigned long long* wc_band = cnt + WC_OFF + (size_t)(round * 2) * WC_PER_LAUNCH;
        unsigned long long* wc_full = wc_band + WC_PER_LAUNCH;
        bool band_used[2] = {false, false};
        // full-matrix launch of one alphabet from per-class segments (lo / cn indexed by sort class) of `lst`
        static const int order[SEG_MAX] = {CLS_FULL, CLS_WIDE0 + 3, CLS_WIDE14 + 3, CLS_WIDE12 + 3, CLS_WIDE10 + 3, CLS_WIDE0 + 2, CLS_WIDE14 + 2, CLS_WIDE12 + 2, CLS_WIDE10 + 2,
                                           CLS_WIDE0 + 1, CLS_WIDE14 + 1, CLS_WIDE12 + 1, CLS_WIDE10 + 1, CLS_WIDE0, CLS_WIDE14, CLS_WIDE12, CLS_WIDE10,
                                           CLS_LANE0 + 4, CLS_LANE0 + 3, CLS_LANE0 + 2, CLS_LANE0 + 1, CLS_LANE0};     // longest serial chains first
        auto launch_fulls = [&](int generic, const long long* lo, const long long* cn_of_in, const uint32_t* lst, hipStream_t fs, unsigned long long* wc, int label, bool skip_blocked = false) -> int {
            const int base = GENERIC_BASE * generic;
            long long cn_local[N_SORT_CLASSES];
            for (int sc = 0; sc < N_SORT_CLASSES; sc++) cn_local[sc] = cn_of_in[sc];
            if (skip_blocked) for (int cls = CLS_WIDE0; cls < N_CLASSES; cls++) if (blocked_words(cls, blocked_kmax)) {      // the pairs behind the split run in row blocks
                long long sp = blocked_split[base + cls]; if (sp < lo[base + cls]) sp = lo[base + cls]; if (sp > lo[base + cls] + cn_local[base + cls]) sp = lo[base + cls] + cn_local[base + cls];
                cn_local[base + cls] = sp - lo[base + cls];
            }
            const long long* cn_of = cn_local;
            FusedTab tf; memset(&tf, 0, sizeof tf);
            unsigned nblk = 0;
            auto class_threads = [&](int cls, long long cn) -> long long {
                if (cls == CLS_FULL) return cn * 64;
                if (cls >= CLS_WIDE0) return cn * (2 << ((cls - CLS_WIDE0) & 3));        // the four wide families: 2 / 4 / 8 / 16 lanes per pair
                return cn;
            };
            // low-latency form of a long class: KIND_LL + words per lane (64 lanes), 0 = none
            auto ll_kind = [](int cls) -> int {
                if (cls == CLS_WIDE0 + 2 || cls == CLS_WIDE12 + 2 || cls == CLS_WIDE14 + 2 || cls == CLS_WIDE10 + 2) return KIND_LL + 2;       // <= 4096 rows
                if (cls == CLS_WIDE12 + 3 || cls == CLS_WIDE10 + 3) return KIND_LL + 3;     // <= 6144
                if (cls == CLS_WIDE0 + 3 || cls == CLS_WIDE14 + 3) return KIND_LL + 4;      // <= 8192
                return 0;
            };
            long long waves_normal = 0, waves_ll = 0;
            for (int k = 0; k < SEG_MAX; k++) {
                const long long cn = cn_of[base + order[k]];
                if (cn <= 0) continue;
                waves_normal += (class_threads(order[k], cn) + 63) / 64;
                waves_ll += ll_kind(order[k]) ? cn : (class_threads(order[k], cn) + 63) / 64;
            }
            static long long ll_waves = 0;                                       // waves per SIMD below which the launch counts as latency-bound
            if (!ll_waves) { const char* e = getenv("SVX_EDIT_LL_WAVES"); ll_waves = e && atoi(e) > 0 ? atoi(e) : 2; }
            const bool low_latency = !getenv("SVX_EDIT_NO_LL") && waves_normal <= ll_waves * 4 * (long long)c->n_cu && waves_ll <= 8 * 4 * (long long)c->n_cu;
            for (int k = 0; k < SEG_MAX; k++) {
                const int cls = order[k];
                const long long cn = cn_of[base + cls];
                if (cn <= 0) continue;
                const int ll = low_latency ? ll_kind(cls) : 0;
                const long long threads = ll ? cn * 64 : class_threads(cls, cn);
                tf.kind[tf.n] = ll ? ll : cls; tf.lo[tf.n] = lo[base + cls]; tf.cn[tf.n] = cn; tf.first_block[tf.n] = nblk;
                nblk += (unsigned)((threads + T - 1) / T); tf.n++;
            }
            tf.first_block[tf.n] = nblk;
            if (!tf.n) return SVX_OK;
            if (serial) HIPCHK(hipEventRecord(c->ev[6], fs));
            if (generic) k_edit_fulls<4><<<nblk, T, 0, fs>>>(tf, lst, scratch, desc, slot_of, ed_dev, cnt + 1, c->e_big_list.as<uint32_t>(), wc);
            else k_edit_fulls<2><<<nblk, T, 0, fs>>>(tf, lst, scratch, desc, slot_of, ed_dev, cnt + 1, c->e_big_list.as<uint32_t>(), wc);
            HIPCHK(hipGetLastError());
            if (serial) {
                HIPCHK(hipEventRecord(c->ev[7], fs));
                HIPCHK(hipStreamSynchronize(fs));
                float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev[6], c->ev[7]));
                fprintf(stderr, "{\"edit_launch\": {\"round\": %d, \"kind\": \"fulls\", \"generic\": %d, \"blocks\": %u, \"ms\": %.4f}}\n", label, generic, nblk, ms);
            }
            return SVX_OK;
        };
        // Round 0 launches its band classes in two parts, widest first: the pairs that fail the WIDEST bands are the long ones whose full matrices are
        // the serial tail of the next round, and they are known as soon as the first part is through - their full-matrix retries start right then,
        // beside the rest of the round (early_cn: what of every retry list has been launched already).
        // the band lists (lo / cn by sort class) of one alphabet as full matrices, one wave per pair (KIND_RETRY)
        auto launch_retry_as_fulls = [&](int generic, const long long* lo, const long long* cn_of, const uint32_t* lst, hipStream_t fs, unsigned long long* wc) -> int {
            const int base = GENERIC_BASE * generic;
            FusedTab tf; memset(&tf, 0, sizeof tf);
            unsigned nblk = 0;
            for (int cls = NBAND - 1; cls >= 0; cls--) {
                const long long cn = cn_of[base + cls];
                if (cn <= 0) continue;
                tf.kind[tf.n] = KIND_RETRY; tf.lo[tf.n] = lo[base + cls]; tf.cn[tf.n] = cn; tf.first_block[tf.n] = nblk;
                nblk += (unsigned)((cn * 64 + T - 1) / T); tf.n++;
            }
            tf.first_block[tf.n] = nblk;
            if (!tf.n) return SVX_OK;
            if (generic) k_edit_fulls<4><<<nblk, T, 0, fs>>>(tf, lst, scratch, desc, slot_of, ed_dev, cnt + 1, c->e_big_list.as<uint32_t>(), wc);
            else k_edit_fulls<2><<<nblk, T, 0, fs>>>(tf, lst, scratch, desc, slot_of, ed_dev, cnt + 1, c->e_big_list.as<uint32_t>(), wc);
            HIPCHK(hipGetLastError());
            return SVX_OK;
        };
        // First class of the early part.  Classes 7, 8 (14 / 16 words) are always in it (the longest pairs; the only ones that need the 16-word kernel).  When classes
        // 5 and 6 hold a real share of the band work (configs[1]: 27 %, the long related pairs; >= 8 % by the proxy pairs x words^2) they join it as a second launch
        // on the early stream (12-word kernel): their failures - full matrices of 3000-5000 rows - then start in the middle of the round instead of at its end
        // (12.7 -> 12.4 ms on configs[1]).  Where they hold next to nothing (the HiFi-like stand-in: 2 %) the extra launch only delays that stream (5.0 -> 6.2 ms there):
        // profiles/r06_edit_split_class_ab.txt.  SVX_EDIT_SPLIT_CLS pins it (7 = classes 7, 8 alone).
        int SPLIT_CLS = 7;
        if (round == 0) {
            double early56 = 0, all = 0;
            for (int generic = 0; generic <= 1; generic++) for (int cls = 0; cls < NBAND; cls++) {
                const double wgt = (double)seg_cn[GENERIC_BASE * generic + cls] * band_words(cls) * band_words(cls);
                all += wgt; if (cls == 5 || cls == 6) early56 += wgt;
            }
            if (all > 0 && early56 >= 0.08 * all) SPLIT_CLS = 5;
            if (const char* e = getenv("SVX_EDIT_SPLIT_CLS")) { const int v = atoi(e); if (v >= 1 && v <= 7) SPLIT_CLS = v; }
        }
        bool split_used[2] = {false, false};
        long long band_pending = 0;
        for (int generic = 0; generic <= 1; generic++) for (int cls = 0; cls < NBAND; cls++) band_pending += seg_cn[GENERIC_BASE * generic + cls];
        const bool bands_as_fulls = round >= 1 && band_pending > 0 && band_pending <= retry_full && !c->edit_force_full;
        for (int generic = 0; generic <= 1; generic++) {
            const int base = GENERIC_BASE * generic;
            if (bands_as_fulls) {
                // the band lists of this alphabet as full matrices, beside the round's full-matrix launch (not behind it)
                SVXCHK(launch_retry_as_fulls(generic, seg_lo, seg_cn, list, full_st[(round + 1) & 1], wc_full));
                SVXCHK(launch_fulls(generic, seg_lo, seg_cn, list, full_st[round & 1], wc_full, round));
                continue;
            }
            bool any_wide = false, any_narrow = false;
            for (int cls = 0; cls < NBAND; cls++) if (seg_cn[base + cls] > 0) { if (cls >= SPLIT_CLS) any_wide = true; else any_narrow = true; }
            const bool split = round == 0 && !serial && any_wide && any_narrow && !getenv("SVX_EDIT_NO_EARLY");
            for (int part3 = 0; part3 < (split ? 3 : 1); part3++) {
                // (split: part3 0 = classes 7, 8; 1 = SPLIT_CLS .. 6, both "part 0" = the early part on its own stream; 2 = the rest)
                const int part = split ? (part3 < 2 ? 0 : 1) : 0;
                FusedTab tb; memset(&tb, 0, sizeof tb);
                tb.narrow = narrow_windows;
                unsigned nblk = 0;
                for (int cls = NBAND - 1; cls >= 0; cls--) {                          // widest band first
                    const long long cn = seg_cn[base + cls];
                    if (cn <= 0) continue;
                    if (split && (part3 == 0 ? cls < 7 : (part3 == 1 ? (cls >= 7 || cls < SPLIT_CLS) : cls >= SPLIT_CLS))) continue;
                    tb.kind[tb.n] = cls; tb.lo[tb.n] = seg_lo[base + cls]; tb.cn[tb.n] = cn; tb.first_block[tb.n] = nblk;
                    nblk += (unsigned)((cn + T - 1) / T); tb.n++;
                }
                tb.first_block[tb.n] = nblk;
                if (!tb.n) continue;
                band_used[generic] = true;
                if (serial) HIPCHK(hipEventRecord(c->ev[6], band_st[generic]));
                // the first part (few, long pairs: latency) on a stream of its own, so that the second does not wait for it
                hipStream_t bs = (split && part == 0) ? early_st[generic] : band_st[generic];
                // the kernel built for 12 state words keeps more waves per SIMD: every launch without the two widest classes takes it
                bool wide16 = false;
                for (int k = 0; k < tb.n; k++) if (band_words(tb.kind[k]) > 12) wide16 = true;
                // (each launch of the early part has lists of its own: everything it failed is launched when IT has ended)
                unsigned long long* const fc = (split && part == 0) ? fail_cnt_early[part3] : fail_cnt;
                uint32_t* const fl = (split && part == 0) ? fb_early[part3] : fb.as<uint32_t>();
                if (generic) {
                    if (wide16) k_edit_bands<4, 16><<<nblk, T, 0, bs>>>(tb, list, scratch, desc, slot_of, ed_dev, fc, fl, pending, wc_band);
                    else k_edit_bands<4, 12><<<nblk, T, 0, bs>>>(tb, list, scratch, desc, slot_of, ed_dev, fc, fl, pending, wc_band);
                } else {
                    if (wide16) k_edit_bands<2, 16><<<nblk, T, 0, bs>>>(tb, list, scratch, desc, slot_of, ed_dev, fc, fl, pending, wc_band);
                    else k_edit_bands<2, 12><<<nblk, T, 0, bs>>>(tb, list, scratch, desc, slot_of, ed_dev, fc, fl, pending, wc_band);
                }
                HIPCHK(hipGetLastError());
                if (split && part == 0) {
                    split_used[generic] = true;
                    MailSrc ms; memset(&ms, 0, sizeof ms); ms.k = 1; ms.p[0] = fail_cnt_early[part3] + base; ms.n[0] = GENERIC_BASE;
                    EarlyPost ep; ep.generic = generic; ep.set = part3;
                    SVXCHK(svx_mail_post(c, bs, ms, &ep.ticket));                     // behind this launch on its stream
                    early_posts.push_back(ep);
                }
                if (serial) {                                                        // SVX_EDIT_SERIAL=1: stand-alone kernel durations for profiling
                    HIPCHK(hipEventRecord(c->ev[7], band_st[generic]));
                    HIPCHK(hipStreamSynchronize(band_st[generic]));
                    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, c->ev[6], c->ev[7]));
                    fprintf(stderr, "{\"edit_launch\": {\"round\": %d, \"kind\": \"bands\", \"generic\": %d, \"blocks\": %u, \"ms\": %.4f}}\n", round, generic, nblk, ms);
                }
            }
            SVXCHK(launch_fulls(generic, seg_lo, seg_cn, list, full_st[round & 1], wc_full, round, round == 0 && blocked_kmax >= 0));
            if (round == 0 && blocked_kmax >= 0) {
                // the row-block route: one launch per block rank on a high-priority stream of its own; boundary words and hand-over values ping-pong between ranks
                FusedTab tb; memset(&tb, 0, sizeof tb);
                BlkTab bt; memset(&bt, 0, sizeof bt);
                unsigned nblk = 0;
                long long slots = 0;
                for (int cls = N_CLASSES - 1; cls >= (int)CLS_WIDE0; cls--) {
                    const int bq = blocked_words(cls, blocked_kmax);
                    if (!bq || seg_cn[base + cls] <= 0) continue;
                    long long sp = blocked_split[base + cls]; if (sp < seg_lo[base + cls]) sp = seg_lo[base + cls]; if (sp > seg_lo[base + cls] + seg_cn[base + cls]) sp = seg_lo[base + cls] + seg_cn[base + cls];
                    const long long cn = seg_lo[base + cls] + seg_cn[base + cls] - sp;
                    if (cn <= 0) continue;
                    tb.kind[tb.n] = bq; tb.lo[tb.n] = sp; tb.cn[tb.n] = cn; tb.first_block[tb.n] = nblk; bt.slot_base[tb.n] = slots;
                    nblk += (unsigned)((cn + T - 1) / T); slots += (cn + 63) / 64 * 64; tb.n++;
                }
                tb.first_block[tb.n] = nblk;
                if (tb.n) {
                    const long long wmax = (long long)((blocked_nmax + 31) / 32) + 1;
                    const size_t bnd_words = (size_t)(slots / 64) * (size_t)wmax * 2 * 64;
                    DevBuf& bb = generic ? c->e_blk[1] : c->e_blk[0];
                    SVXCHK(bb.reserve((bnd_words * 2 + (size_t)slots * 2) * 4 + 256));
                    uint32_t* bnd0 = bb.as<uint32_t>(); uint32_t* bnd1 = bnd0 + bnd_words;
                    int* e0 = reinterpret_cast<int*>(bnd1 + bnd_words); int* e1 = e0 + slots;
                    hipStream_t bs = c->aux[7];
                    const int ranks = 2 << blocked_kmax;
                    for (int r = 0; r < ranks; r++) {
                        BlkArgs A; A.rank = r; A.wmax = wmax;
                        A.bnd_in = (r & 1) ? bnd0 : bnd1; A.bnd_out = (r & 1) ? bnd1 : bnd0; A.e_in = (r & 1) ? e0 : e1; A.e_out = (r & 1) ? e1 : e0;
                        if (generic) k_edit_blocked<4><<<nblk, T, 0, bs>>>(tb, bt, A, list, scratch, desc, slot_of, ed_dev, wc_full);
                        else k_edit_blocked<2><<<nblk, T, 0, bs>>>(tb, bt, A, list, scratch, desc, slot_of, ed_dev, wc_full);
                    }
                    HIPCHK(hipGetLastError());
                    blocked_used = true;
                }
            }
        }
        // the early parts, in the order they end: everything such a launch failed - full-matrix classes in their forms, band classes (class 4..6 can fail into class 8)
        // as one-wave-per-pair full matrices - starts now, beside the rest of the round
        for (int set = 0; set < 2; set++) for (const EarlyPost& ep : early_posts) {
            if (ep.set != set) continue;
            const unsigned long long* w = nullptr;
            SVXCHK(svx_mail_wait(c, early_st[ep.generic], ep.ticket, &w));
            long long e_lo[N_SORT_CLASSES], e_cn[N_SORT_CLASSES];
            for (int sc = 0; sc < N_SORT_CLASSES; sc++) { e_lo[sc] = (long long)sc * pending; e_cn[sc] = 0; }
            for (int k = 0; k < GENERIC_BASE; k++) e_cn[GENERIC_BASE * ep.generic + k] = (long long)w[k];
            hipStream_t fs = full_st[(round + 1) & 1];
            unsigned long long* wc_next = cnt + WC_OFF + (size_t)((round + 1) * 2 + 1) * WC_PER_LAUNCH;
            SVXCHK(launch_fulls(ep.generic, e_lo, e_cn, fb_early[ep.set], fs, wc_next, round + 1));
            SVXCHK(launch_retry_as_fulls(ep.generic, e_lo, e_cn, fb_early[ep.set], fs, wc_next));
        }
        // only the band launches can hand pairs to the next round
        const long long cap = pending;
        pending = 0;
        for (int sc = 0; sc < N_SORT_CLASSES; sc++) { seg_lo[sc] = (long long)sc * cap; seg_cn[sc] = 0; }
        if (band_used[0] || band_used[1]) {
            // one post for the round: the main stream (idle during the rounds, not a low-priority one) waits for every band stream and sends the counters
            for (int g = 0; g <= 1; g++) if (band_used[g]) { HIPCHK(hipEventRecord(c->ev[22 + g], band_st[g])); HIPCHK(hipStreamWaitEvent(st, c->ev[22 + g], 0)); }
            for (int g = 0; g <= 1; g++) if (split_used[g]) { HIPCHK(hipEventRecord(c->ev[20 + g], early_st[g])); HIPCHK(hipStreamWaitEvent(st, c->ev[20 + g], 0)); }
            unsigned long long h[N_SORT_CLASSES];
            SVXCHK(svx_mail_read(c, st, fail_cnt, N_SORT_CLASSES, h));
            for (int sc = 0; sc < N_SORT_CLASSES; sc++) {
                seg_cn[sc] = (long long)h[sc];                                    // (the first part's failures have lists of their own and are running already)
                pending += seg_cn[sc];
            }
        }
        list = fb.as<uint32_t>();
    }
    // the end of the rounds: the main stream waits for both full-matrix streams, sums the counters (k_edit_tail) and posts them - one wait for the host
    unsigned long long tail[6];
    for (int k = 0; k < 2; k++) { HIPCHK(hipEventRecord(c->ev[22 + k], full_st[k])); HIPCHK(hipStreamWaitEvent(st, c->ev[22 + k], 0)); }
    if (blocked_used) { HIPCHK(hipEventRecord(c->ev[19], c->aux[7])); HIPCHK(hipStreamWaitEvent(st, c->ev[19], 0)); }
    k_edit_tail<<<1, 256, 0, st>>>(cnt, cnt + WC_OFF, MAX_ROUNDS, cnt + 8);             // (the class bounds at cnt[8..] are history by now)
    HIPCHK(hipGetLastError());
    SVXCHK(svx_mail_read(c, st, cnt + 8, 6, tail));
    guess = __builtin_bit_cast(float, (uint32_t)tail[1]);
    c->edit_guess_last = guess;
    if (profile && !first_desc.empty()) {
        // how much wider than necessary was the first band of every pair?  (needed = narrowest band class that certifies the distance found in the end)
        std::vector<long long> slot((size_t)n_work);
        long long n_slot = n_work;
        if (slot_of) {
            SVXCHK(svx_d2h(slot.data(), slot_of, (size_t)n_work * 8, st));
            for (long long w = 0; w < n_work; w++) if (slot[(size_t)w] + 1 > n_slot) n_slot = slot[(size_t)w] + 1;
        }
        std::vector<int32_t> ed((size_t)n_slot);
        SVXCHK(svx_d2h(ed.data(), ed_dev, (size_t)n_slot * 4, st));
        double used[N_CLASSES] = {0}, need[N_CLASSES] = {0}, failed[N_CLASSES] = {0}; long long cnt_c[N_CLASSES] = {0};
        double ratio_hist[8] = {0};
        auto h_need_class = [](int m, int n, int d) { int x = (d - (n - m)) / 2; if (x < 0) x = 0; const int nw = (n - m) + 2 * x + 1;
            for (int b = 0; b < NBAND; b++) if (nw + (b < FIRST_STAIR_CLS ? 14 : 46) <= 32 * band_words(b)) return b; return (int)CLS_FULL; };
        for (long long w = 0; w < n_work; w++) {
            const PairDesc& pd = first_desc[(size_t)w];
            if (pd.cls == -1) continue;
            const int cls = pd.cls & 0xff;
            if (cls >= NBAND) continue;
            const int d = ed[(size_t)(slot_of ? slot[(size_t)w] : w)];
            const int nc = h_need_class(pd.m, pd.n, d);
            const double u = (double)pd.n * band_words(cls), nd = (double)pd.n * (nc < NBAND ? band_words(nc) : (pd.m + 31) / 32);
            cnt_c[cls]++; used[cls] += u;
            if (nc <= cls) { need[cls] += nd; int r = band_words(cls) * 4 / band_words(nc) - 4; if (r > 7) r = 7; ratio_hist[r] += u; }
            else failed[cls] += u;
        }
        // The full matrices of round 0 (pairs whose length gap exceeds every band): how many of their word-columns (32 rows x 1 column) lie outside the diagonals
        // [-x, (n - m) + x] a tile decomposition could skip?  x from the upper bound the pair came with (what such a kernel would know beforehand) and from the
        // distance found in the end (perfect foresight).  A word of rows [32 w, 32 w + 31] is needed in the columns 32 w - x .. 32 w + 31 + (n - m) + x
        for (int grp = 0; grp < 2; grp++) {                                          // 0: whole pattern in one lane (<= 512 rows), 1: the multi-lane forms
            double all = 0, by_ub = 0, by_d = 0; long long pairs = 0;
            for (long long w = 0; w < n_work; w++) {
                const PairDesc& pd = first_desc[(size_t)w];
                if (pd.cls == -1) continue;
                const int cls = pd.cls & 0xff;
                if (cls < NBAND || (cls >= (int)CLS_LANE0 && cls < (int)CLS_WIDE0) != (grp == 0)) continue;
                const int d = ed[(size_t)(slot_of ? slot[(size_t)w] : w)], m = pd.m, n = pd.n, gap = n - m;
                if (gap < 0) continue;
                pairs++;
                for (int which = 0; which < 2; which++) {
                    const int bound = which == 0 ? (pd.ub < m + n ? pd.ub : m + n) : d;
                    const int x = bound > gap ? (bound - gap + 1) / 2 : 0;
                    double cols = 0;
                    for (int r0 = 0; r0 < m; r0 += 32) {
                        int lo = r0 - x, hi = r0 + 31 + gap + x;
                        if (lo < 0) lo = 0;
                        if (hi > n - 1) hi = n - 1;
                        cols += hi >= lo ? hi - lo + 1 : 0;
                    }
                    (which == 0 ? by_ub : by_d) += cols;
                }
                all += (double)n * ((m + 31) / 32);
            }
            fprintf(stderr, "{\"edit_full_matrix_corners\": {\"form\": \"%s\", \"pairs\": %lld, \"word_cols\": %.4g, \"inside_band_of_upper_bound\": %.4g, \"inside_band_of_final_distance\": %.4g, "
                            "\"skippable_fraction_known_beforehand\": %.4f, \"skippable_fraction_perfect_foresight\": %.4f}}\n",
                    grp == 0 ? "one lane per pair" : "multi-lane", pairs, all, by_ub, by_d, all > 0 ? 1.0 - by_ub / all : 0.0, all > 0 ? 1.0 - by_d / all : 0.0);
        }
        for (int b = 0; b < NBAND; b++) if (cnt_c[b])
            fprintf(stderr, "{\"edit_band_fit\": {\"first_cls\": %d, \"words\": %d, \"pairs\": %lld, \"word_cols_first_band\": %.4g, \"of_which_failed\": %.4g, \"word_cols_narrowest_sufficient\": %.4g}}\n",
                    b, band_words(b), cnt_c[b], used[b], failed[b], need[b]);
        fprintf(stderr, "{\"edit_band_fit_ratio_hist\": {\"bins\": \"first band / narrowest sufficient band in [1, 1.25), [1.25, 1.5) ... >= 2.75, weighted by word-columns\", \"h\": [%.4g, %.4g, %.4g, %.4g, %.4g, %.4g, %.4g, %.4g]}}\n",
                ratio_hist[0], ratio_hist[1], ratio_hist[2], ratio_hist[3], ratio_hist[4], ratio_hist[5], ratio_hist[6], ratio_hist[7]);
    }
    {
        const unsigned long long nb = tail[0];
        c->stats.n_edit_wordcols_issued += (int64_t)tail[2]; c->stats.n_edit_wordcols_useful += (int64_t)tail[3];
        c->stats.n_edit_wordcols_retry += (int64_t)tail[4]; c->stats.n_edit_wordcols_band += (int64_t)tail[5];
        c->stats.edit_guess = guess;
        if (nb) {
            // rare: shorter core > 16384 symbols.  Size the block-state scratch exactly from the core lengths.
            const long long nbig = (long long)nb;
            std::vector<uint32_t> items((size_t)nbig);
            SVXCHK(svx_d2h(items.data(), c->e_big_list.p, (size_t)nbig * 4, st));
            std::vector<PairDesc> pd((size_t)nbig);
            { HostCopy hc(st); for (long long i = 0; i < nbig; i++) SVXCHK(hc.d2h(&pd[(size_t)i], desc + items[(size_t)i], sizeof(PairDesc))); SVXCHK(hc.finish()); }
            std::vector<long long> off((size_t)nbig + 1, 0);
            for (long long i = 0; i < nbig; i++) off[(size_t)i + 1] = off[(size_t)i] + (((long long)pd[(size_t)i].m + 31) / 32) * 7;
            SVXCHK(c->e_big_state.reserve((size_t)off[(size_t)nbig] * 4 + 16));
            SVXCHK(c->e_big_off.reserve((size_t)(nbig + 1) * 8));
            SVXCHK(svx_h2d(c->e_big_off.p, off.data(), (size_t)(nbig + 1) * 8, st));
            k_edit_full_big<<<(unsigned)nbig, 64, 0, st>>>(nbig, c->e_big_list.as<uint32_t>(), c->e_big_off.as<long long>(), scratch, desc, slot_of, ed_dev,
                                                          c->e_big_state.as<uint32_t>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    if (profile && n_work >= 4096) {
        // the divergence histogram of the exact distances, next to what the pilot guessed from its sample
        SVXCHK(c->e_hist.reserve(256 * 8));
        HIPCHK(hipMemsetAsync(c->e_hist.p, 0, 256 * 8, st));
        k_edit_hist<<<(unsigned)(c->n_cu * 2), 256, 0, st>>>(n_work, desc, slot_of, ed_dev, c->e_hist.as<unsigned long long>());
        unsigned long long h[256];
        HIPCHK(hipMemcpyAsync(h, c->e_hist.p, sizeof h, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        double total = 0, acc = 0;
        for (int b = 0; b < 256; b++) total += (double)h[b];
        fprintf(stderr, "{\"edit_guess_used\": %.4f, \"edit_guess_exact_histogram\": %.4f, \"cum_weight_by_bin\": [", guess, guess_from_histogram(h, guess));
        for (int b = 0; b < 256; b++) { acc += (double)h[b]; if (b % 8 == 7) fprintf(stderr, "%.3f%s", total > 0 ? acc / total : 0.0, b == 255 ? "" : ", "); }
        fprintf(stderr, "]}\n");
    }
    return SVX_OK;
}

int svx_edit_distance_pairs(svx_ctx* c, int64_t n_pairs, const uint8_t* codes_dev, const int64_t* a_off_dev, const int64_t* b_off_dev,
                            int32_t* out_dev) {
    PairSource src; memset(&src, 0, sizeof src);
    src.plain = 1; src.codes = codes_dev; src.a_off = a_off_dev; src.b_off = b_off_dev; src.g_codes = codes_dev;
    return run_edit_pipeline(c, n_pairs, src, out_dev, nullptr);
}

// The packed store of a CLUSTER call does not depend on the pair list except through the flank radius, and every pair that needs an edit
// distance has |start_a - start_b| <= 2 * cluster_max_distance * position_distance_normalizer (ins_needs_edit in cluster.hip): with that bound as
// radius the store is built on a side stream while the main stream sorts, partitions and samples.  begin: word counts + offsets (enqueue only);
// pack: called after the caller's next host synchronisation - reads the total, reserves, packs, records ev[18].
// The store is built on the high-priority stream of the full-matrix rounds (idle at that time): on the HiFi-like stand-in k_edit_prep waits for it, and at the
// main stream's priority or below it arrives 0.6-0.9 ms later (profiles/r06_prepack_priority_ab.txt).  What the high priority cost configs[1] - the main stream's
// one-workgroup scans of the partition sizes waited 0.04-0.4 ms each for a CU with sixteen free wave slots while the store's blocks were being dispatched - is
// taken care of in scan.hpp (256-thread workgroups for small scans).  SVX_PREPACK_PRIO=normal / low: A/B.
static hipStream_t prepack_stream(svx_ctx* c) {
    static const int which = []() { const char* e = getenv("SVX_PREPACK_PRIO"); return e && !strcmp(e, "normal") ? 6 : (e && !strcmp(e, "low") ? 0 : 2); }();
    return c->aux[which];
}

static PairSource prepack_source(svx_ctx* c, const ClusterIn& in) {
    PairSource src; memset(&src, 0, sizeof src);
    src.plain = 0; src.in = in; src.g_off = c->g_off_p; src.g_codes = c->g_codes_p; src.radius = c->prepack_radius; src.rec = c->e_rec.as<HapRec>();
    return src;
}

int svx_edit_prepack_begin(svx_ctx* c, const ClusterIn& in, const svx_params& p, hipEvent_t input_ready) {
    c->prepack_state = 0;
    if (!c->g_off_p || !c->pinned || in.n <= 0 || getenv("SVX_EDIT_NO_PREPACK")) return SVX_OK;
    const double bound = 2.0 * p.cluster_max_distance * p.position_distance_normalizer;
    if (!(bound >= 0) || !(bound < 16000.0)) return SVX_OK;                       // unusual parameters: the exact radius is found from the pair list
    c->prepack_radius = (long long)(bound * (1.0 + 1e-9)) + 2 + 100;
    c->prepack_n = in.n;
    hipStream_t ps = prepack_stream(c);
    const long long n_rec = in.n;
    SVXCHK(c->e_words.reserve((size_t)(n_rec + 1) * 8));
    SVXCHK(c->e_off.reserve((size_t)(n_rec + 1) * 8));
    SVXCHK(c->e_rec.reserve((size_t)n_rec * sizeof(HapRec) + 64));
    const PairSource src = prepack_source(c, in);
    HIPCHK(hipStreamWaitEvent(ps, input_ready, 0));
    k_hap_words<<<(unsigned)((n_rec + 1 + 255) / 256), 256, 0, ps>>>(n_rec, src, c->e_words.as<int64_t>());
    SVXCHK(svx_exclusive_scan_i64_on(c, c->e_words.as<int64_t>(), c->e_off.as<int64_t>(), n_rec + 1, ps, c->prepack_tmp));
    HIPCHK(hipMemcpyAsync(c->pinned, c->e_off.as<int64_t>() + n_rec, 8, hipMemcpyDeviceToHost, ps));
    c->prepack_state = 1;
    return SVX_OK;
}

int svx_edit_prepack_pack(svx_ctx* c, const ClusterIn& in) {
    if (c->prepack_state != 1) return SVX_OK;
    hipStream_t ps = prepack_stream(c);
    HIPCHK(hipStreamSynchronize(ps));
    const int64_t total_words = c->pinned[0];
    SVXCHK(c->e_scratch.reserve((size_t)(total_words + 64) * 4));
    const long long n_rec = in.n;
    const PairSource src = prepack_source(c, in);
    HIPCHK(hipMemsetAsync(c->e_scratch.p, 0, 4, ps));                    // the leading pad word
    k_hap_pack<<<(unsigned)((n_rec + 3) / 4), 256, 0, ps>>>(n_rec, src, c->e_off.as<int64_t>(), c->e_scratch.as<uint32_t>(), c->e_rec.as<HapRec>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[18], ps));
    c->prepack_state = 2;
    return SVX_OK;
}

// used by cluster.hip
int svx_launch_edit_pairs(svx_ctx* c, int64_t n_work, const void* work_dev, const ClusterIn& in, int32_t* ed_dev,
                          unsigned long long* cells_dev) {
    PairSource src; memset(&src, 0, sizeof src);
    src.plain = 0; src.work = (const EditWork*)work_dev; src.in = in; src.g_off = c->g_off_p; src.g_codes = c->g_codes_p;
    return run_edit_pipeline(c, n_work, src, ed_dev, cells_dev);
}

// loads this translation unit's code object (HIP does it lazily, at the first launch): called by svx_ctx_create so that the first COLLECT / CLUSTER call
// of a context does not pay for it
void svx_preload_edit() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_edit_classify)); (void)hipGetLastError(); }
