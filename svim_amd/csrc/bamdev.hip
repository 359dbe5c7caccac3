// bamdev.hip - the BAM front-end with everything after the file read ON THE DEVICE:
//   BGZF blocks --H2D--> k_bgzf_inflate --> the inflated BAM stream in HBM (it never crosses PCIe)
//   --> record discovery (k_anchor / k_walk: BGZF blocks are the restart points of the block_size chain)
//   --> k_measure / k_fields / k_cigar_copy / k_name_*: fixed fields, packed CIGAR (incl. the CG long-CIGAR tag), the SA tag parsed into the segment
//       table, read names interned by 2 x 64-bit hashes - the svx_batch of include/svx.h with DEVICE pointers; SEQ is not copied at all (svx_batch.seq
//       points into the inflated stream, seq_off at each record's packed bases).
// Replaces, for coordinate-sorted BAM input: pysam.AlignmentFile(bam).fetch(until_eof=True) + the per-record accessors + the SA string handling of
// src/svim/SVIM_COLLECT.py:44-93,132-167 (what bamio.cpp's decode_run / append_sa do on the host's cores).
#include "common.hpp"
#include "hostcopy.hpp"
#include "devdec.hpp"
#include "scan.hpp"
#include <zlib.h>
#include <chrono>
#include <mutex>
#include <cstdlib>
#include <thread>

#define DD_NONE (~0ull)
#define DD_HEAD ((size_t)32 << 20)            // room in front of a chunk's inflated data for the unconsumed tail of the chunk before
#define DD_MAX_REC (1u << 28)
#define DD_HBUF ((size_t)2048 << 20)           // pinned host memory for the host cores' share of a chunk's inflate
#define DD_E_CORRUPT 1
#define DD_E_AUX 2
#define DD_E_SA 3
#define DD_E_HASH 4
#define DD_E_CHAIN 5

static inline double dd_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- unaligned little-endian reads from the inflated stream (records start at any byte; the buffer is 256-byte aligned and padded) ----------------
__device__ __forceinline__ uint32_t ld32(const uint8_t* st, uint64_t p) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(st) + (p >> 2);
    const uint32_t sh = (uint32_t)(p & 3u) * 8u;
    const uint32_t lo = w[0];
    return sh ? (lo >> sh) | (w[1] << (32u - sh)) : lo;
}

__host__ __device__ inline uint64_t dd_fnv(const uint8_t* p, uint32_t n, uint64_t seed) {
    uint64_t h = 0xcbf29ce484222325ull ^ seed;
    for (uint32_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}

// could a BAM record start at byte p?  (SAM spec 4.2: block_size, refID, pos, l_read_name/mapq/bin, n_cigar_op/flag, l_seq, next_refID, next_pos, tlen)
__device__ __forceinline__ bool dd_plausible(const uint8_t* st, uint64_t p, uint64_t data_end, int32_t n_ref, const int32_t* ref_len) {
    if (p + 36 > data_end) return false;
    const uint32_t bs = ld32(st, p);
    if (bs < 32u || bs > DD_MAX_REC) return false;
    const int32_t tid = (int32_t)ld32(st, p + 4);
    if (tid < -1 || tid >= n_ref) return false;
    const int32_t pos = (int32_t)ld32(st, p + 8);
    if (pos < -1 || (tid >= 0 && pos > ref_len[tid])) return false;
    const uint32_t w3 = ld32(st, p + 12), w4 = ld32(st, p + 16);
    const uint32_t l_name = w3 & 0xffu, n_cig = w4 & 0xffffu;
    const int32_t l_seq = (int32_t)ld32(st, p + 20);
    if (l_name == 0u || l_seq < 0) return false;
    const int32_t ntid = (int32_t)ld32(st, p + 24), npos = (int32_t)ld32(st, p + 28);
    if (ntid < -1 || ntid >= n_ref || npos < -1) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    const uint64_t nul = p + 36 + l_name - 1;
    if (nul < data_end && st[nul] != 0) return false;
    return true;
}
// follow the chain for a few records: every one must look like a record, its first CIGAR operations must be operations
__device__ bool dd_validate(const uint8_t* st, uint64_t q, uint64_t data_end, int32_t n_ref, const int32_t* ref_len) {
    for (int hop = 0; hop < 3; hop++) {
        if (q + 36 > data_end) return true;
        if (!dd_plausible(st, q, data_end, n_ref, ref_len)) return false;
        const uint32_t bs = ld32(st, q), l_name = ld32(st, q + 12) & 0xffu, n_cig = ld32(st, q + 16) & 0xffffu;
        const uint64_t cig = q + 36 + l_name;
        for (uint32_t j = 0; j < n_cig && j < 4u; j++) { if (cig + 4 * j + 4 > data_end) break; if ((ld32(st, cig + 4 * j) & 15u) > 8u) return false; }
        q += 4ull + bs;
    }
    return true;
}

// ---- CRC32 of every inflated block against its BGZF trailer (htslib verifies it in bgzf_read_block; SVX_BAM_VERIFY_CRC=0 switches the check off) ---------------
// One wavefront per block, 4 KiB per round: lane l runs the table-driven byte recurrence over its 64 bytes of the round from register 0 (the lanes of a
// wave read one contiguous 4 KiB), the 64 registers are combined pairwise in six levels - register(A || B) = shift(register(A), |B|) ^ register(B), the
// shift by 2^j zero bytes being a fixed 32 x 32 matrix over GF(2) - and the rounds are chained the same way.  The block is right-aligned in its rounds
// (leading zero bytes leave a zero register alone); the initial value 0xffffffff enters at the end as shift(0xffffffff, len).
struct CrcJob { unsigned long long at; uint32_t len, crc; };
#define CRC_POW 17                                  /* shift matrices for 2^0 .. 2^16 zero bytes */
typedef uint32_t __attribute__((aligned(1))) crc_u32_unaligned;
__device__ __forceinline__ uint32_t crc_apply(const uint32_t* __restrict__ m, uint32_t x) {       // m: 32 words, wave-uniform address
    uint32_t o = 0;
#pragma unroll
    for (int bit = 0; bit < 32; bit++) o ^= m[bit] & (0u - ((x >> bit) & 1u));
    return o;
}
__global__ __launch_bounds__(64) void k_crc32(const uint8_t* st, const CrcJob* jobs, long long nb, const uint32_t* __restrict__ shift, int* err) {
    __shared__ uint32_t T[256], T1[256], T2[256], T3[256];        // T: one byte; T1..T3: the same byte followed by 1..3 zero bytes (a word takes one round of look-ups)
    const int lane = lane_id();
    for (int e = lane; e < 256; e += 64) {
        uint32_t c = (uint32_t)e;
#pragma unroll
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        T[e] = c;
    }
    __syncthreads();
    for (int e = lane; e < 256; e += 64) {
        const uint32_t c1 = T[T[e] & 255u] ^ (T[e] >> 8), c2 = T[c1 & 255u] ^ (c1 >> 8), c3 = T[c2 & 255u] ^ (c2 >> 8);
        T1[e] = c1; T2[e] = c2; T3[e] = c3;
    }
    __syncthreads();
    const long long b = blockIdx.x;
    if (b >= nb) return;
    const CrcJob job = jobs[b];
    const uint8_t* base = st + job.at;
    const long long len = job.len, rounds = (len + 4095) >> 12, pad = (rounds << 12) - len;
    uint32_t acc = 0;
    for (long long r = 0; r < rounds; r++) {
        const long long off = (r << 12) + (long long)lane * 64 - pad;            // where my 64 bytes start in the block (negative: virtual zero bytes)
        uint32_t reg = 0;
        if (off >= 0) {
            const crc_u32_unaligned* w = reinterpret_cast<const crc_u32_unaligned*>(base + off);
            uint32_t x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = w[k];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                reg ^= x[k];
                reg = T3[reg & 255u] ^ T2[(reg >> 8) & 255u] ^ T1[(reg >> 16) & 255u] ^ T[reg >> 24];
            }
        } else if (off > -64) {
            for (long long i = 0; i < off + 64; i++) reg = T[(reg ^ base[i]) & 255u] ^ (reg >> 8);
        }
        // six levels: the last lane of every group of 2, 4, ... 64 holds the register of its group's bytes
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t left = (uint32_t)__shfl_up((int)reg, 1 << j, 64);
            const uint32_t joined = crc_apply(shift + 32 * (6 + j), left) ^ reg;
            if ((lane & ((2 << j) - 1)) == (2 << j) - 1) reg = joined;
        }
        acc = crc_apply(shift + 32 * 12, acc) ^ (uint32_t)__builtin_amdgcn_readlane((int)reg, 63);
    }
    uint32_t init = 0xffffffffu;
    for (int j = 0; j < CRC_POW; j++) if ((len >> j) & 1) init = crc_apply(shift + 32 * j, init);
    const uint32_t crc = ~(acc ^ init);
    if (lane == 0 && crc != job.crc) { if (atomicCAS(err, 0, 1) == 0) err[1] = (int)b; }
}
// the shift matrices: column `bit` of matrix j = the register that 1 << bit becomes after 2^j zero bytes
static void crc_shift_matrices(uint32_t (*m)[32]) {
    uint32_t T[256];
    for (uint32_t e = 0; e < 256; e++) { uint32_t c = e; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; T[e] = c; }
    for (int bit = 0; bit < 32; bit++) { const uint32_t r = 1u << bit; m[0][bit] = T[r & 255u] ^ (r >> 8); }
    for (int j = 1; j < CRC_POW; j++)
        for (int bit = 0; bit < 32; bit++) { uint32_t o = 0; for (int k = 0; k < 32; k++) if ((m[j - 1][bit] >> k) & 1u) o ^= m[j - 1][k]; m[j][bit] = o; }
}

// first record start at or after the start of every BGZF block (one wavefront per block; lanes test 64 consecutive byte offsets at a time)
__global__ __launch_bounds__(64) void k_anchor(const uint8_t* st, const uint64_t* blk_off, long long nb, uint64_t data_end, int32_t n_ref, const int32_t* ref_len,
                                                uint64_t* anchor) {
    const long long b = blockIdx.x;
    if (b >= nb) return;
    const int lane = lane_id();
    const uint64_t lo = blk_off[b], hi = blk_off[b + 1];
    for (uint64_t base = lo; base < hi; base += 64) {
        const uint64_t p = base + (uint64_t)lane;
        unsigned long long m = __ballot(p < hi && dd_plausible(st, p, data_end, n_ref, ref_len));
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            if (dd_validate(st, base + (uint64_t)k, data_end, n_ref, ref_len)) { if (lane == 0) anchor[b] = base + (uint64_t)k; return; }
        }
    }
    if (lane == 0) anchor[b] = DD_NONE;
}

// the records that start inside block b, walked from its anchor (thread per block).  base == NULL: count only; exit_at[b] = where the walk left the block
__global__ void k_walk(const uint8_t* st, const uint64_t* blk_off, long long nb, uint64_t data_end, const uint64_t* anchor, uint32_t* cnt, uint64_t* exit_at,
                       const uint64_t* base, uint64_t* rec_off) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint64_t p = anchor[b];
    if (p == DD_NONE) { if (!base) { cnt[b] = 0; exit_at[b] = DD_NONE; } return; }
    const uint64_t hi = blk_off[b + 1];
    uint32_t n = 0;
    const uint64_t at = base ? base[b] : 0;
    while (p < hi) {
        if (p + 4 > data_end) break;
        const uint64_t nx = p + 4ull + ld32(st, p);
        if (nx > data_end) break;                                    // the chunk ends inside this record: it is the tail the next chunk starts with
        if (base) rec_off[at + n] = p;
        n++;
        p = nx;
    }
    if (!base) { cnt[b] = n; exit_at[b] = p; }
}
// fallback when the anchors do not link up (never seen; kept so that the result does not depend on a heuristic): one thread follows the whole chain
__global__ void k_walk_serial(const uint8_t* st, uint64_t begin, uint64_t data_end, uint64_t cap, uint64_t* rec_off, unsigned long long* n_out, uint64_t* tail) {
    if (blockIdx.x || threadIdx.x) return;
    uint64_t p = begin, n = 0;
    while (p + 4 <= data_end) {
        const uint64_t nx = p + 4ull + ld32(st, p);
        if (nx > data_end) break;
        if (rec_off && n < cap) rec_off[n] = p;
        n++;
        p = nx;
    }
    *n_out = n; *tail = p;
}

struct RecDesc { uint64_t cig_at, sa_at, seq_at, h1, h2; uint32_t n_cig, sa_len, name_len, flags; };      // flags: 1 = eligible primary with an SA tag

__device__ __forceinline__ bool dd_digit(uint8_t c) { return c >= '0' && c <= '9'; }

// per record (one thread): sizes - effective CIGAR (CG tag), SA string and what it will expand to, name hashes
__global__ void k_measure(const uint8_t* st, const uint64_t* rec_off, long long n, int min_mapq, RecDesc* desc, uint32_t* n_cig, uint32_t* n_seg, uint32_t* n_segop,
                          int* err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = rec_off[i], r = p + 4, end = r + ld32(st, p);
    const uint32_t w = ld32(st, r + 8), w2 = ld32(st, r + 12);
    const uint32_t l_name = w & 0xffu, mapq = (w >> 8) & 0xffu, ncf = w2 & 0xffffu, flag = w2 >> 16;
    const uint32_t l_seq = ld32(st, r + 16);
    RecDesc d;
    d.cig_at = r + 32 + l_name; d.n_cig = ncf; d.seq_at = d.cig_at + 4ull * ncf; d.sa_at = 0; d.sa_len = 0; d.flags = 0;
    d.name_len = l_name ? l_name - 1u : 0u;
    if (end - r < 32 || d.seq_at + ((uint64_t)l_seq + 1) / 2 + l_seq > end) { atomicExch(err, DD_E_CORRUPT); d.n_cig = 0; d.name_len = 0; }
    d.h1 = dd_fnv(st + r + 32, d.name_len, 0) | 1ull;                  // 0 marks an empty slot
    d.h2 = dd_fnv(st + r + 32, d.name_len, 0x9E3779B97F4A7C15ull);
    // htslib's bam_tag2cigar (what pysam hands the reference): a mapped record (tid, pos >= 0) whose FIRST operation is a soft clip of the whole read carries
    // its real CIGAR in CG:B,I (or B,i) - whatever the rest of the placeholder looks like
    const bool want_cg = ncf >= 1u && (int32_t)ld32(st, r) >= 0 && (int32_t)ld32(st, r + 4) >= 0 && d.n_cig == ncf && (ld32(st, d.cig_at) & 15u) == 4u && (ld32(st, d.cig_at) >> 4) == l_seq;
    const bool primary_ok = !(flag & (4u | 256u | 2048u)) && (int)mapq >= min_mapq;
    uint32_t segs = 0, ops = 0;
    if ((want_cg || primary_ok) && d.n_cig == ncf) {
        uint64_t q = d.seq_at + ((uint64_t)l_seq + 1) / 2 + l_seq;
        uint64_t cg_at = 0; uint32_t cg_n = 0; bool have_cg = false, cg_seen = false, sa_seen = false;      // (bam_aux_get: the FIRST field named CG counts, whatever its type)
        while (q + 3 <= end) {
            const uint8_t t0 = st[q], t1 = st[q + 1], ty = st[q + 2];
            q += 3;
            const uint64_t left = end - q;
            uint64_t sz = 0;
            if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
            else if (ty == 's' || ty == 'S') sz = 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
            else if (ty == 'Z' || ty == 'H') {
                uint64_t z = q;
                while (z < end && st[z] != 0) z++;
                if (z >= end) { atomicExch(err, DD_E_AUX); break; }
                if (t0 == 'S' && t1 == 'A' && ty == 'Z' && !sa_seen) { d.sa_at = q; d.sa_len = (uint32_t)(z - q); }
                sz = z - q + 1;
            } else if (ty == 'B') {
                if (left < 5) { atomicExch(err, DD_E_AUX); break; }
                const uint8_t sub = st[q];
                const uint32_t c = ld32(st, q + 1);
                const uint64_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
                sz = 5 + es * (uint64_t)c;
                if (sz <= left && t0 == 'C' && t1 == 'G' && !cg_seen && (sub == 'I' || sub == 'i') && c > 0u) { cg_at = q + 5; cg_n = c; have_cg = true; }
            } else { atomicExch(err, DD_E_AUX); break; }
            if (t0 == 'C' && t1 == 'G') cg_seen = true;
            if (t0 == 'S' && t1 == 'A') sa_seen = true;
            if (sz > left) { atomicExch(err, DD_E_AUX); break; }
            q += sz;
        }
        if (want_cg && have_cg && cg_n >= d.n_cig && cg_n < (1u << 29)) { d.cig_at = cg_at; d.n_cig = cg_n; }      // (only an array at least as long as the placeholder)      // long CIGARs (> 65535 operations) live in CG:B,I behind a <l_seq>S<ref_len>N placeholder
        if (primary_ok && d.sa_len) {
            // what the SA string expands to: entries with exactly 6 fields (src/svim/SVIM_COLLECT.py:55-62), operations of their CIGAR field
            d.flags = 1u;
            uint64_t a = d.sa_at; const uint64_t e = d.sa_at + d.sa_len;
            while (a < e) {
                uint64_t z = a; uint32_t commas = 0, o = 0, flen = 0; bool star = false;
                for (; z < e && st[z] != ';'; z++) {
                    const uint8_t ch = st[z];
                    if (ch == ',') commas++;
                    else if (commas == 3u) { flen++; if (ch == '*') star = true; else if (!dd_digit(ch)) o++; }
                }
                if (z > a && commas == 5u) { segs++; ops += (star && flen == 1u) ? 0u : o + (star ? 1u : 0u); }      // "*" alone = no operations (k_fields: the same test); a '*' inside a CIGAR is a bad operation there
                a = z + 1;
            }
        }
    }
    desc[i] = d;
    n_cig[i] = d.n_cig; n_seg[i] = segs; n_segop[i] = ops;
}

struct ContigTable { const uint64_t* key; const int32_t* tid; uint32_t mask; const char* names; const uint32_t* name_off; };

__device__ bool dd_parse_int(const uint8_t* s, uint64_t n, long long* v) {              // bamio.cpp parse_int
    if (n == 0) return false;
    uint64_t i = 0; bool neg = false;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; if (n == 1) return false; }
    long long x = 0;
    for (; i < n; i++) { if (!dd_digit(s[i])) return false; x = x * 10 + (s[i] - '0'); if (x > (1ll << 40)) return false; }
    *v = neg ? -x : x;
    return true;
}

// per record (one thread): the fixed fields of the batch, and the SA tag expanded into segment rows (bamio.cpp append_sa;
// src/svim/SVIM_COLLECT.py:55-85: 6-field check, pos - 1, strand, mapq overflow -> 0)
__global__ void k_fields(const uint8_t* st, const uint64_t* rec_off, long long n, const RecDesc* desc, const uint32_t* seg_off, const uint64_t* segop_off,
                         ContigTable ct, uint16_t* flag, int32_t* tid, int32_t* pos, uint8_t* mapq, int32_t* lseq, uint64_t* seq_off, int32_t* seg_tid,
                         int32_t* seg_pos, uint8_t* seg_rev, uint8_t* seg_mapq, int32_t* seg_lseq, uint64_t* seg_cigar_off, uint32_t* seg_cigar, int* err,
                         unsigned long long* n_warn) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t r = rec_off[i] + 4;
    const RecDesc d = desc[i];
    const uint32_t w = ld32(st, r + 8), w2 = ld32(st, r + 12);
    const int32_t l_seq = (int32_t)ld32(st, r + 16);
    tid[i] = (int32_t)ld32(st, r); pos[i] = (int32_t)ld32(st, r + 4); mapq[i] = (uint8_t)((w >> 8) & 0xffu); lseq[i] = l_seq;
    flag[i] = (uint16_t)(((w2 >> 16) & 0x0fffu) | (d.flags & 1u ? SVX_FLAG_SA : 0u));
    seq_off[i] = d.seq_at;
    if (!(d.flags & 1u)) return;
    uint32_t row = seg_off[i];
    uint64_t cw = segop_off[i];
    uint64_t a = d.sa_at; const uint64_t e = d.sa_at + d.sa_len;
    while (a < e) {
        uint64_t z = a, fs[7]; int nf = 0; fs[0] = a;
        for (; z < e && st[z] != ';'; z++) if (st[z] == ',') { if (nf < 6) fs[++nf] = z + 1; else nf++; }
        if (z > a) {
            if (nf == 5) {
                fs[6] = z + 1;
                long long p1 = 0, mq = 0, nm = 0;
                if (!dd_parse_int(st + fs[1], fs[2] - 1 - fs[1], &p1) || !dd_parse_int(st + fs[4], fs[5] - 1 - fs[4], &mq) || !dd_parse_int(st + fs[5], fs[6] - 1 - fs[5], &nm)) {
                    atomicExch(err, DD_E_SA); return;
                }
                if (mq < 0 || mq > 255) mq = 0;
                // reference name -> id (open addressing over the header's names; an unknown name yields -1)
                const uint32_t ln = (uint32_t)(fs[1] - 1 - fs[0]);
                const uint64_t hk = dd_fnv(st + fs[0], ln, 0) | 1ull;
                int32_t t = -1;
                for (uint32_t s = (uint32_t)hk & ct.mask;; s = (s + 1u) & ct.mask) {
                    const uint64_t k = ct.key[s];
                    if (!k) break;
                    if (k == hk) {
                        const int32_t cand = ct.tid[s];
                        const uint32_t o0 = ct.name_off[cand], o1 = ct.name_off[cand + 1];
                        bool same = o1 - o0 == ln;
                        for (uint32_t c = 0; same && c < ln; c++) same = (uint8_t)ct.names[o0 + c] == st[fs[0] + c];
                        if (same) { t = cand; break; }
                    }
                }
                seg_tid[row] = t; seg_pos[row] = (int32_t)(p1 - 1);
                seg_rev[row] = (fs[3] - 1 - fs[2] == 1 && st[fs[2]] == '+') ? 0 : 1;
                seg_mapq[row] = (uint8_t)mq; seg_lseq[row] = l_seq;
                const uint64_t c0 = fs[3], c1 = fs[4] - 1;
                if (!(c1 - c0 == 1 && st[c0] == '*')) {
                    long long num = 0; bool have = false;
                    for (uint64_t c = c0; c < c1; c++) {
                        const uint8_t ch = st[c];
                        if (dd_digit(ch)) { num = num * 10 + (ch - '0'); have = true; }
                        else {
                            int op = -1;
                            const char* ops = "MIDNSHP=XB";
                            for (int k = 0; k < 10; k++) if ((uint8_t)ops[k] == ch) op = k;
                            if (op < 0 || !have) { atomicExch(err, DD_E_SA); return; }
                            seg_cigar[cw++] = (uint32_t)(num << 4) | (uint32_t)op;
                            num = 0; have = false;
                        }
                    }
                    if (have) { atomicExch(err, DD_E_SA); return; }
                }
                row++;
                seg_cigar_off[row] = cw;
            } else atomicAdd(n_warn, 1ull);
        }
        a = z + 1;
    }
}

// packed CIGAR words of every record -> one aligned array (one wavefront per record; the words sit at any byte offset in the stream)
__global__ __launch_bounds__(256) void k_cigar_copy(const uint8_t* st, long long n, const RecDesc* desc, const uint64_t* cigar_off, uint32_t* cigar) {
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = lane_id();
    const uint64_t src = desc[i].cig_at, dst = cigar_off[i];
    const uint32_t m = desc[i].n_cig;
    for (uint32_t j = (uint32_t)lane; j < m; j += 64u) cigar[dst + j] = ld32(st, src + 4ull * j);
}

// ---- read names -> dense ids (equal names get equal ids across the whole file; SVIM_clustering.py:141-167 compares the reads of signatures) --------
// key = 64-bit hash (open addressing, atomicCAS), verified by a second, independent 64-bit hash: two different names would have to agree in 128 bits
struct NameTable { unsigned long long* key; unsigned long long* check; int32_t* id; uint32_t mask; };

__global__ void k_name_insert(long long n, const RecDesc* desc, NameTable nt, int32_t id_base, uint32_t* slot_of, unsigned int* n_new, uint32_t* new_rec) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long h1 = desc[i].h1;
    uint32_t s = (uint32_t)(h1 >> 1) & nt.mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&nt.key[s], 0ull, h1);
        if (old == 0ull) {                                             // first record of this name: it numbers the read
            const unsigned int k = atomicAdd(n_new, 1u);
            new_rec[k] = (uint32_t)i;
            nt.check[s] = desc[i].h2;
            nt.id[s] = id_base + (int32_t)k;
            break;
        }
        if (old == h1) break;
        s = (s + 1u) & nt.mask;
    }
    slot_of[i] = s;
}
__global__ void k_name_ids(long long n, const RecDesc* desc, NameTable nt, const uint32_t* slot_of, int32_t* read_id, int* err) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = slot_of[i];
    if (nt.check[s] != desc[i].h2) atomicExch(err, DD_E_HASH);
    read_id[i] = nt.id[s];
}
__global__ void k_name_lens(long long n_new, const uint32_t* new_rec, const RecDesc* desc, uint32_t* len) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_new) len[k] = desc[new_rec[k]].name_len + 1u;          // with its NUL
}
__global__ void k_name_copy(const uint8_t* st, const uint64_t* rec_off, long long n_new, const uint32_t* new_rec, const RecDesc* desc, const uint64_t* at, char* blob) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_new) return;
    const uint64_t src = rec_off[new_rec[k]] + 36;
    const uint32_t m = desc[new_rec[k]].name_len;
    char* dst = blob + at[k];
    for (uint32_t c = 0; c < m; c++) dst[c] = (char)st[src + c];
    dst[m] = 0;
}
__global__ void k_name_rehash(uint32_t old_cap, const unsigned long long* okey, const unsigned long long* ocheck, const int32_t* oid, NameTable nt) {
    const uint32_t s0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (s0 >= old_cap || !okey[s0]) return;
    const unsigned long long h1 = okey[s0];
    uint32_t s = (uint32_t)(h1 >> 1) & nt.mask;
    while (atomicCAS(&nt.key[s], 0ull, h1) != 0ull) s = (s + 1u) & nt.mask;
    nt.check[s] = ocheck[s0]; nt.id[s] = oid[s0];
}

__global__ void k_first_beyond(long long n, const int32_t* tid, int32_t limit, unsigned long long* first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (tid[i] < 0 || tid[i] > limit)) atomicMin(first, (unsigned long long)i);
}
__global__ void k_order_iota(long long n, uint32_t* order, uint32_t* seg_order) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { order[i] = (uint32_t)(2 * i); seg_order[i] = (uint32_t)(2 * i + 1); }
}
__global__ void k_widen_u32(long long n, const uint32_t* in, uint64_t* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// ---- query-name-sorted input (src/svim/SVIM_COLLECT.py:8-41 bam_iterator, :96-129 analyze_alignment_file_querysorted; bamio.cpp svx_bam_read_batch mode 1) ----
// Consecutive records with one read name are a group.  A group is analysed iff it holds exactly ONE record that is neither secondary nor supplementary and that
// one is mapped with mapq >= min_mapq (:108); its good supplementary records (mapped, mapq >= min_mapq, :112) are analysed too, and THEY - not the SA tag - are the
// segment rows of the primary.  Everything else carries SVX_FLAG_SKIP.  Emission slots per analysed read: primary, good supplementaries.., segments (:114-121).
// Groups are small (a handful of records), so nothing here walks a group: group index and ranks inside a group come from scans over the batch.
// first record index of the last group in [0, n) (the group the chunk may end in the middle of)
__global__ void k_q_last_group(long long n, const int32_t* read_id, unsigned long long* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (i == 0 || read_id[i] != read_id[i - 1])) atomicMax(out, (unsigned long long)i);
}
// first group boundary at or behind `from` (n when the records from `from` on are one group with the record before)
__global__ void k_q_next_boundary(long long from, long long n, const int32_t* read_id, unsigned long long* out) {
    const long long i = from + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && i > 0 && read_id[i] != read_id[i - 1]) atomicMin(out, (unsigned long long)i);
}
// per record of the batch: group head?  good supplementary?  (0 / 1 as u32 for the scans)
__global__ void k_q_marks(long long n, const uint16_t* flag, const uint8_t* mapq, const int32_t* read_id, int min_mapq, uint32_t* head, uint32_t* good) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { head[i] = 0u; good[i] = 0u; return; }
    const unsigned f = flag[i];
    head[i] = (i == 0 || read_id[i] != read_id[i - 1]) ? 1u : 0u;
    good[i] = (!(f & 256u) && (f & 2048u) && !(f & 4u) && (int)mapq[i] >= min_mapq) ? 1u : 0u;
}
// group g starts at record gs[g]; primaries counted per group, the (last) primary's index kept
__global__ void k_q_groups(long long n, const uint32_t* head, const uint32_t* gidx_ex, const uint16_t* flag, uint32_t* gs, uint32_t* n_prim, uint32_t* p_idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = gidx_ex[i] + head[i] - 1u;                  // exclusive scan of head + own head - 1 = index of the record's group
    if (head[i]) gs[g] = (uint32_t)i;
    const unsigned f = flag[i];
    if (!(f & 256u) && !(f & 2048u)) { atomicAdd(n_prim + g, 1u); atomicMax(p_idx + g, (uint32_t)i); }
}
// per group: analysed?  its slots and segment rows
__global__ void k_q_verdict(long long n_groups, long long n, const uint32_t* gs, const uint32_t* n_prim, const uint32_t* p_idx, const uint16_t* flag, const uint8_t* mapq,
                            int min_mapq, const uint32_t* good_ex, uint32_t* slots, uint32_t* rows) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g > n_groups) return;
    if (g == n_groups) { slots[g] = 0u; rows[g] = 0u; return; }
    const uint32_t a = gs[g], b = g + 1 < n_groups ? gs[g + 1] : (uint32_t)n;
    bool ok = n_prim[g] == 1u;
    if (ok) { const uint32_t p = p_idx[g]; ok = !(flag[p] & 4u) && (int)mapq[p] >= min_mapq; }
    const uint32_t n_good = ok ? good_ex[b] - good_ex[a] : 0u;
    slots[g] = ok ? n_good + 2u : 0u;
    rows[g] = n_good;
}
// per record: flag with SVX_FLAG_SKIP, emission slots, segment offsets; a good supplementary of an analysed read also fills its segment row
__global__ void k_q_records(long long n, long long n_groups, const uint32_t* head, const uint32_t* gidx_ex, const uint32_t* gs, const uint32_t* p_idx, const uint32_t* good,
                            const uint32_t* good_ex, const uint32_t* slots, const uint32_t* slot_ex, const uint32_t* row_ex, const uint16_t* flag, const int32_t* tid,
                            const int32_t* pos, const uint8_t* mapq, const int32_t* lseq, const uint64_t* cigar_off, uint16_t* flag_out, uint32_t* order, uint32_t* seg_order,
                            uint32_t* seg_off, int32_t* seg_tid, int32_t* seg_pos, uint8_t* seg_rev, uint8_t* seg_mapq, int32_t* seg_lseq, uint32_t* row_rec, uint32_t* row_ops) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { seg_off[i] = row_ex[n_groups]; return; }
    const uint32_t g = gidx_ex[i] + head[i] - 1u;
    const bool ok = slots[g] != 0u;
    const uint32_t n_good = ok ? slots[g] - 2u : 0u;
    const uint32_t p = p_idx[g];
    uint16_t f = (uint16_t)((flag[i] & 0x0fffu) | SVX_FLAG_SKIP);
    uint32_t o = 0u, so = 0u;
    seg_off[i] = row_ex[g] + ((ok && (uint32_t)i > p) ? n_good : 0u);         // the rows belong to the primary: records behind it start past them
    if (ok) {
        if ((uint32_t)i == p) { f &= (uint16_t)~SVX_FLAG_SKIP; o = slot_ex[g]; so = slot_ex[g] + 1u + n_good; }
        else if (good[i]) {
            const uint32_t q = good_ex[i] - good_ex[gs[g]];
            f &= (uint16_t)~SVX_FLAG_SKIP; o = slot_ex[g] + 1u + q;
            const uint32_t r = row_ex[g] + q;
            seg_tid[r] = tid[i]; seg_pos[r] = pos[i]; seg_rev[r] = (flag[i] & 16u) ? 1 : 0; seg_mapq[r] = mapq[i]; seg_lseq[r] = lseq[i];
            row_rec[r] = (uint32_t)i; row_ops[r] = (uint32_t)(cigar_off[i + 1] - cigar_off[i]);
        }
    }
    flag_out[i] = f; order[i] = o; seg_order[i] = so;
}
// CIGAR words of the segment rows: one wave per row
__global__ __launch_bounds__(256) void k_q_seg_cigar(long long n_rows, const uint32_t* row_rec, const uint64_t* cigar_off, const uint32_t* cigar, const uint64_t* seg_cigar_off,
                                                     uint32_t* seg_cigar) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const uint64_t src = cigar_off[row_rec[r]], dst = seg_cigar_off[r], cnt = seg_cigar_off[r + 1] - dst;
    for (uint64_t k = threadIdx.x & 63; k < cnt; k += 64) seg_cigar[dst + k] = cigar[src + k];
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
struct DevChunk {
    DevBuf stream; size_t data_begin = 0, data_end = 0, tail_start = 0;            // offsets into stream
    uint64_t seq_end_host = 0;
    DevBuf blk_off, anchor, cnt, exit_at, base, rec_off, desc, n_cig, n_seg, n_segop, scan_tmp, crc_jobs;
    std::vector<CrcJob> crc_host;          // (alive until the chunk is loaded again: its upload is asynchronous)
    DevBuf flag, tid, pos, mapq, lseq, read_id, cigar_off, cigar, seq_off, seg_off, segop_off, seg_tid, seg_pos, seg_rev, seg_mapq, seg_lseq, seg_cigar_off, seg_cigar;
    DevBuf slot_of, new_rec, name_len, name_at, name_blob;
    DevBuf order[2], seg_order[2]; int order_flip = 0;
    // query-name mode: per handed-out batch (two alternating sets like order[]) the flags with SVX_FLAG_SKIP, the segment table made of the good supplementary records
    DevBuf q_flag[2], q_seg_off[2], q_seg_tid[2], q_seg_pos[2], q_seg_rev[2], q_seg_mapq[2], q_seg_lseq[2], q_seg_cigar_off[2], q_seg_cigar[2];
    DevBuf q_head, q_good, q_gidx, q_good_ex, q_gs, q_nprim, q_pidx, q_slots, q_rows, q_slot_ex, q_row_ex, q_row_rec, q_row_ops;
    int64_t n_rec = 0, tot_seg = 0, tot_ops = 0, tot_segops = 0;
    bool loaded = false;
    void release() {
        DevBuf* all[] = {&stream, &blk_off, &anchor, &cnt, &exit_at, &base, &rec_off, &desc, &n_cig, &n_seg, &n_segop, &scan_tmp, &crc_jobs, &flag, &tid, &pos, &mapq, &lseq, &read_id,
                         &cigar_off, &cigar, &seq_off, &seg_off, &segop_off, &seg_tid, &seg_pos, &seg_rev, &seg_mapq, &seg_lseq, &seg_cigar_off, &seg_cigar, &slot_of,
                         &new_rec, &name_len, &name_at, &name_blob, &order[0], &order[1], &seg_order[0], &seg_order[1],
                         &q_flag[0], &q_flag[1], &q_seg_off[0], &q_seg_off[1], &q_seg_tid[0], &q_seg_tid[1], &q_seg_pos[0], &q_seg_pos[1], &q_seg_rev[0], &q_seg_rev[1],
                         &q_seg_mapq[0], &q_seg_mapq[1], &q_seg_lseq[0], &q_seg_lseq[1], &q_seg_cigar_off[0], &q_seg_cigar_off[1], &q_seg_cigar[0], &q_seg_cigar[1],
                         &q_head, &q_good, &q_gidx, &q_good_ex, &q_gs, &q_nprim, &q_pidx, &q_slots, &q_rows, &q_slot_ex, &q_row_ex, &q_row_rec, &q_row_ops};
        for (auto* b : all) b->release();
    }
};

struct svx_devdec {
    int device = 0;
    hipStream_t stream = nullptr;              // the chunk loader's (devdec_load / devdec_count: a background thread of the reader)
    hipStream_t batch_stream = nullptr;        // the consumer's (devdec_batch): a chunk it hands out is complete, nothing of it waits behind the NEXT chunk's inflate
    svx_inflater* inf = nullptr;
    int32_t n_ref = 0;
    DevBuf ref_len, contig_rank, ct_key, ct_tid, ct_names, ct_name_off, err, counters, crc_shift, batch_cnt;
    uint32_t ct_mask = 0;
    DevBuf nt_key, nt_check, nt_id; uint32_t nt_cap = 0;
    std::vector<std::string> names;
    DevChunk chunk[3];
    DevDecStats stats;
    int n_threads = 8; hipStream_t copy_stream = nullptr; uint8_t* hbuf = nullptr; size_t hbuf_cap = 0;      // the host's share of the inflate
    int* h_err = nullptr;                      // pinned (DD_PINNED_BYTES: 64 bytes for h_err, then the DD_H_* values of h_cnt)
    unsigned long long* h_cnt = nullptr;       // pinned: indexed by the DD_H_* slots below, each written by one copy and read after the synchronise that follows it
};

// slots of svx_devdec::h_cnt (pinned read-backs).  The loader thread (devdec_load / devdec_count) and the consumer (devdec_batch) use disjoint slots.
enum { DD_H_OPS = 0, DD_H_WALK_TAIL = 1, DD_H_SEGOPS = 2, DD_H_SEG = 4, DD_H_SA_BAD = 6, DD_H_NEW = 8, DD_H_BLOB = 10, DD_H_FIRST_BEYOND = 12, DD_H_TAIL_BS = 13,
       DD_H_LAST_GROUP = 16, DD_H_LAST_OFF = 17, DD_H_NEXT_BOUNDARY = 18, DD_H_GROUPS = 20, DD_H_ROWS = 21, DD_H_ROW_OPS = 22, DD_H_SLOTS = 24 };
#define DD_PINNED_BYTES 320
static_assert(64 + DD_H_SLOTS * sizeof(unsigned long long) <= DD_PINNED_BYTES, "the pinned block of svx_devdec holds h_err (64 bytes) and DD_H_SLOTS values of h_cnt");

static int dd_alloc_names(svx_devdec* d, uint32_t cap) {
    SVXCHK(d->nt_key.reserve((size_t)cap * 8)); SVXCHK(d->nt_check.reserve((size_t)cap * 8)); SVXCHK(d->nt_id.reserve((size_t)cap * 4));
    HIPCHK(hipMemsetAsync(d->nt_key.p, 0, (size_t)cap * 8, d->stream));
    d->nt_cap = cap;
    return SVX_OK;
}

int devdec_create(int device, int n_threads, int32_t n_ref, const int32_t* ref_len, const char* names_blob, const int32_t* contig_rank, svx_devdec** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return svx_fail(SVX_E_NODEVICE, "no such GPU (the device BAM decode has no CPU fallback)", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(device));
    svx_devdec* d = new svx_devdec();
    d->device = device; d->n_ref = n_ref;
    d->n_threads = n_threads > 0 ? n_threads : 1;
    HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&d->batch_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
    SVXCHK(svx_inflater_create(device, &d->inf));
    { void* p = nullptr; HIPCHK(hipHostMalloc(&p, DD_PINNED_BYTES, hipHostMallocDefault)); memset(p, 0, DD_PINNED_BYTES); d->h_err = (int*)p; d->h_cnt = (unsigned long long*)((char*)p + 64); }
    SVXCHK(d->err.reserve(64)); SVXCHK(d->counters.reserve(256));
    const size_t nr = (size_t)(n_ref > 0 ? n_ref : 1);
    SVXCHK(d->ref_len.reserve(nr * 4)); SVXCHK(d->contig_rank.reserve(nr * 4));
    if (n_ref > 0) {
        SVXCHK(svx_h2d(d->ref_len.p, ref_len, nr * 4, d->stream));
        SVXCHK(svx_h2d(d->contig_rank.p, contig_rank, nr * 4, d->stream));
    }
    // reference names: hash table name -> id for the SA tags
    uint32_t cap = 16; while (cap < 4u * (uint32_t)nr) cap <<= 1;
    std::vector<uint64_t> key(cap, 0); std::vector<int32_t> tid(cap, -1); std::vector<uint32_t> off(nr + 1, 0);
    std::string blob;
    const char* p = names_blob;
    for (int32_t t = 0; t < n_ref; t++) {
        const size_t ln = strlen(p);
        off[(size_t)t] = (uint32_t)blob.size(); blob.append(p, ln);
        const uint64_t h = dd_fnv(reinterpret_cast<const uint8_t*>(p), (uint32_t)ln, 0) | 1ull;
        uint32_t s = (uint32_t)h & (cap - 1);
        while (key[s]) s = (s + 1) & (cap - 1);
        key[s] = h; tid[s] = t;
        p += ln + 1;
    }
    off[nr > (size_t)n_ref ? (size_t)n_ref : nr] = (uint32_t)blob.size();
    if (n_ref > 0) off[(size_t)n_ref] = (uint32_t)blob.size();
    SVXCHK(d->ct_key.reserve((size_t)cap * 8)); SVXCHK(d->ct_tid.reserve((size_t)cap * 4)); SVXCHK(d->ct_names.reserve(blob.size() + 16)); SVXCHK(d->ct_name_off.reserve((nr + 1) * 4));
    SVXCHK(svx_h2d(d->ct_key.p, key.data(), (size_t)cap * 8, d->stream));
    SVXCHK(svx_h2d(d->ct_tid.p, tid.data(), (size_t)cap * 4, d->stream));
    SVXCHK(svx_h2d(d->ct_names.p, blob.data(), blob.size(), d->stream));
    SVXCHK(svx_h2d(d->ct_name_off.p, off.data(), (nr + 1) * 4, d->stream));
    d->ct_mask = cap - 1;
    SVXCHK(dd_alloc_names(d, 1u << 20));
    HIPCHK(hipStreamSynchronize(d->stream));
    *out = d;
    return SVX_OK;
}

void devdec_destroy(svx_devdec* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipStreamSynchronize(d->stream);
    (void)hipStreamSynchronize(d->batch_stream);
    if (d->inf) svx_inflater_destroy(d->inf);
    for (auto& c : d->chunk) c.release();
    DevBuf* all[] = {&d->ref_len, &d->contig_rank, &d->ct_key, &d->ct_tid, &d->ct_names, &d->ct_name_off, &d->err, &d->counters, &d->crc_shift, &d->batch_cnt, &d->nt_key, &d->nt_check, &d->nt_id};
    for (auto* b : all) b->release();
    if (d->h_err) (void)hipHostFree(d->h_err);
    if (d->hbuf) (void)hipHostFree(d->hbuf);
    (void)hipStreamDestroy(d->copy_stream);
    (void)hipStreamDestroy(d->batch_stream);
    (void)hipStreamDestroy(d->stream);
    delete d;
}

const std::vector<std::string>& devdec_names(svx_devdec* d) { return d->names; }
void devdec_stats(svx_devdec* d, DevDecStats* out) { *out = d->stats; }
void devdec_reset_names(svx_devdec* d) { (void)d; }

template <class T> static int dd_scan(svx_devdec* d, DevChunk& c, const T* in, T* out, size_t n, hipStream_t st = nullptr) {           // exclusive; out[n] NOT written
    return svx_exclusive_scan<T, T>(in, out, (long long)n, st ? st : d->stream, c.scan_tmp);
}

static int dd_check(svx_devdec* d, const char* where) {
    HIPCHK(hipMemcpyAsync(d->h_err, d->err.p, 4, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    const int e = *d->h_err;
    if (!e) return SVX_OK;
    const char* what = e == DD_E_CORRUPT ? "corrupt BAM record" : e == DD_E_AUX ? "malformed BAM aux field" : e == DD_E_SA ? "malformed SA tag" :
                       e == DD_E_HASH ? "two different read names share both 64-bit hashes" : "record chain";
    char msg[160];
    snprintf(msg, sizeof msg, "device BAM decode (%s): %s", where, what);
    return svx_fail(SVX_E_ARG, msg, __FILE__, __LINE__, hipSuccess);
}

#define GRIDB(n, t) (unsigned)(((n) + (t) - 1) / (t))

int devdec_load(svx_devdec* d, int slot, const DevDecBlock* blocks, size_t nb_in, int carry_slot, uint64_t skip_bytes, bool final_chunk, int min_mapq, int mode) {
    if (mode == 1) min_mapq = 1000;            // query-name mode: the segment rows of a read are its supplementary RECORDS (devdec_batch), no SA tag is expanded
    if (!d || slot < 0 || slot > 2) return svx_fail(SVX_E_ARG, "bad slot", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(d->device));
    hipStream_t st = d->stream;
    DevChunk& c = d->chunk[slot];
    c.loaded = false; c.n_rec = c.tot_seg = c.tot_ops = c.tot_segops = 0;
    // ---- layout of the inflated data -------------------------------------------------------------------------------------------------------------
    std::vector<uint64_t> out_at(nb_in + 1, 0);
    for (size_t k = 0; k < nb_in; k++) out_at[k + 1] = out_at[k] + blocks[k].isize;
    const uint64_t total = out_at[nb_in];
    size_t carry = 0;
    const DevChunk* prev = carry_slot >= 0 ? &d->chunk[carry_slot] : nullptr;
    if (prev && prev->loaded) carry = prev->data_end - prev->tail_start;
    if (carry > DD_HEAD) return svx_fail(SVX_E_CAPACITY, "BAM record larger than the carry-over room of the device reader", __FILE__, __LINE__, hipSuccess);
    SVXCHK(c.stream.reserve(DD_HEAD + (size_t)total + 256));
    uint8_t* sp = c.stream.as<uint8_t>();
    if (carry) HIPCHK(hipMemcpyAsync(sp + DD_HEAD - carry, prev->stream.as<uint8_t>() + prev->tail_start, carry, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(sp + DD_HEAD + total, 0, 192, st));
    c.data_begin = DD_HEAD - carry + (carry ? 0 : (size_t)skip_bytes);
    c.data_end = DD_HEAD + (size_t)total;
    HIPCHK(hipStreamSynchronize(st));
    // ---- inflate: the GPU takes sub-batches of blocks from the FRONT of the chunk (file slice -> pinned staging -> H2D -> k_bgzf_inflate straight into
    // the stream, three sub-batches in flight), the host's cores take runs of blocks from the BACK with zlib into pinned memory, uploaded into their
    // place; whoever is faster takes more (SVX_BAM_DEV_CPU=0: GPU only) --------------------------------------------------------------------------------
    double t0 = dd_now();
    {
        const size_t sub_env = []() { const char* e = getenv("SVX_BAM_DEV_SUB"); return e && atoll(e) > 0 ? (size_t)atoll(e) : (size_t)0; }();       // (experiments)
        const int cpu_env = []() { const char* e = getenv("SVX_BAM_DEV_CPU"); return e ? atoi(e) : -1; }();
        // the lane-per-block decoder (SVX_INFLATE_LANES=1) wants ~64 k blocks in flight whatever the launch sizes are (a launch takes ~80 ms from 1 to 64 k blocks):
        // three sub-batches of a third of that, no ramp.  (Six rolling sub-batches of 11 k blocks measured worse - 0.82 against 0.96 M records/s on the 8 GB chunk,
        // profiles/r06_end_to_end_lane_decoder.txt: more streams than hardware queues.)
        const bool lanes = []() { const char* e = getenv("SVX_INFLATE_LANES"); return e && e[0] == '1'; }();
        const int NS = []() { const char* e = getenv("SVX_BAM_DEV_SLOTS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 8 ? v : 3; }();      // sub-batches in flight
        const size_t SUB = sub_env ? sub_env : (lanes ? 21845 : 32768);   // wave-per-block: a launch of >= ~25 k blocks runs at the kernel's best rate (a 12 k one at 0.8 of it: its tail)
        int n_cpu = cpu_env >= 0 ? cpu_env : (d->n_threads > 8 ? d->n_threads - 6 : (d->n_threads > 3 ? d->n_threads - 3 : 0));      // (the staging copies want cores, too)
        if (nb_in < 4 * SUB / 3) n_cpu = 0;                            // a small chunk: one launch does it
        std::mutex m;
        size_t lo = 0, hi = nb_in;
        size_t h_first_v = 0;                                              // (set below, before any worker runs)
        auto take = [&](bool front, size_t want, size_t& a, size_t& b) -> bool {
            std::lock_guard<std::mutex> g(m);
            if (lo >= hi) return false;
            if (front) { a = lo; b = hi < lo + want ? hi : lo + want; lo = b; }
            else {
                const size_t floor_ = lo > h_first_v ? lo : h_first_v;
                if (hi <= floor_) return false;
                b = hi; a = hi > floor_ + want ? hi - want : floor_; hi = a;
            }
            return true;
        };
        std::vector<std::string> errs((size_t)(n_cpu > 0 ? n_cpu : 1));
        std::vector<int64_t> cpu_done((size_t)(n_cpu > 0 ? n_cpu : 1), 0);
        std::vector<std::thread> workers;
        // the host's share lands in pinned memory first: room for the BACK of the chunk (the cores take a quarter or so; at most DD_HBUF of it)
        const size_t want_h = (size_t)total < DD_HBUF ? (size_t)total : DD_HBUF;
        if (n_cpu > 0 && d->hbuf_cap < want_h) {
            if (d->hbuf) (void)hipHostFree(d->hbuf);
            d->hbuf = nullptr; d->hbuf_cap = 0;
            void* p = nullptr;
            if (hipHostMalloc(&p, want_h + 4096, hipHostMallocDefault) == hipSuccess) { d->hbuf = (uint8_t*)p; d->hbuf_cap = want_h; }
            else { (void)hipGetLastError(); n_cpu = 0; }
        }
        const uint64_t h_base = total > d->hbuf_cap ? total - d->hbuf_cap : 0;          // inflated offsets at or above this may go through the host buffer
        size_t h_first = 0;                                                               // first block the cores may take
        while (h_first < nb_in && out_at[h_first] < h_base) h_first++;
        h_first_v = h_first;
        for (int w = 0; w < n_cpu; w++) workers.emplace_back([&, w]() {
            (void)hipSetDevice(d->device);
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { errs[(size_t)w] = "inflateInit2 failed"; return; }
            size_t a, b;
            while (take(false, 16, a, b)) {
                for (size_t k = a; k < b; k++) {
                    if (!blocks[k].isize) continue;
                    if (inflateReset(&zs) != Z_OK) { errs[(size_t)w] = "inflateReset failed"; break; }
                    zs.next_in = const_cast<Bytef*>(blocks[k].comp); zs.avail_in = blocks[k].clen;
                    zs.next_out = d->hbuf + (out_at[k] - h_base); zs.avail_out = blocks[k].isize;
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END || zs.avail_out != 0) { errs[(size_t)w] = "BGZF inflate failed (host share of the device reader)"; break; }
                }
                if (!errs[(size_t)w].empty()) break;
                if (out_at[b] > out_at[a] && hipMemcpyAsync(sp + DD_HEAD + out_at[a], d->hbuf + (out_at[a] - h_base), (size_t)(out_at[b] - out_at[a]), hipMemcpyHostToDevice, d->copy_stream) != hipSuccess) {
                    errs[(size_t)w] = "upload of host-inflated blocks failed"; break;
                }
                cpu_done[(size_t)w] += (int64_t)(b - a);
            }
            inflateEnd(&zs);
        });
        int rc_gpu = SVX_OK;
        {
            bool used[8] = {false, false, false, false, false, false, false, false};
            int sl = 0;
            std::vector<uint64_t> in_off, o_at; std::vector<uint32_t> clen, isz;
            size_t a, b;
            // sub-batch sizes: small first ones (the GPU starts after 4 k blocks are staged), then SUB
            size_t ramp = (sub_env || lanes) ? SUB : 4096;
            for (;;) {
                const size_t want = ramp;
                if (!(rc_gpu == SVX_OK && want > 0 && take(true, want, a, b))) break;
                if (ramp < SUB) ramp *= 2;
                const size_t mm = b - a;
                if (used[sl]) { float ms = 0; rc_gpu = svx_inflater_wait(d->inf, sl, &ms); d->stats.inflate_kernel_ms += ms; used[sl] = false; if (rc_gpu != SVX_OK) break; }
                const uint8_t* f0 = blocks[a].comp;
                const uint64_t staged = (uint64_t)(blocks[b - 1].comp + blocks[b - 1].clen - f0);
                in_off.resize(mm); o_at.resize(mm); clen.resize(mm); isz.resize(mm);
                for (size_t k = 0; k < mm; k++) { in_off[k] = (uint64_t)(blocks[a + k].comp - f0); clen[k] = blocks[a + k].clen; isz[k] = blocks[a + k].isize; o_at[k] = out_at[a + k] - out_at[a]; }
                {
                    uint8_t* stage = (uint8_t*)svx_inflater_staging(d->inf, sl, staged + 8);
                    if (!stage) { rc_gpu = svx_fail(SVX_E_HIP, "no pinned staging memory", __FILE__, __LINE__, hipSuccess); break; }
                    {   // the slice of the file as it is, copied by a few threads (page cache -> pinned memory)
                        const int parts = staged > ((uint64_t)4 << 20) ? (n_cpu > 0 ? 4 : 8) : 1;
                        std::vector<std::thread> cp;
                        for (int q = 1; q < parts; q++) {
                            const uint64_t l0 = staged * (uint64_t)q / parts, h0 = staged * (uint64_t)(q + 1) / parts;
                            cp.emplace_back([=]() { memcpy(stage + l0, f0 + l0, (size_t)(h0 - l0)); });
                        }
                        memcpy(stage, f0, (size_t)(staged / parts));
                        for (auto& t : cp) t.join();
                    }
                    rc_gpu = svx_inflater_enqueue(d->inf, sl, (int64_t)mm, in_off.data(), clen.data(), isz.data(), o_at.data(), staged, sp + DD_HEAD + out_at[a], out_at[b] - out_at[a], 1);
                }
                if (rc_gpu != SVX_OK) break;
                used[sl] = true; d->stats.gpu_blocks += (int64_t)mm;
                sl = (sl + 1) % NS;
            }
            d->stats.t_stage += dd_now() - t0; t0 = dd_now();
            for (int k = 0; k < NS; k++) if (used[k]) { float ms = 0; const int rc = svx_inflater_wait(d->inf, k, &ms); d->stats.inflate_kernel_ms += ms; if (rc_gpu == SVX_OK) rc_gpu = rc; }
        }
        if (rc_gpu != SVX_OK) { std::lock_guard<std::mutex> g(m); lo = hi; }        // let the workers run out
        for (auto& t : workers) t.join();
        if (n_cpu > 0) HIPCHK(hipStreamSynchronize(d->copy_stream));
        d->stats.t_inflate_wait += dd_now() - t0; t0 = dd_now();
        if (rc_gpu != SVX_OK) return rc_gpu;
        for (auto& e : errs) if (!e.empty()) return svx_fail(SVX_E_ARG, e.c_str(), __FILE__, __LINE__, hipSuccess);
        for (auto v : cpu_done) d->stats.cpu_blocks += v;
    }
    d->stats.blocks += (int64_t)nb_in; d->stats.bytes += (int64_t)total;
    c.tail_start = c.data_begin;
    if (c.data_end <= c.data_begin) { c.loaded = true; c.tail_start = c.data_begin < c.data_end ? c.data_begin : c.data_end; return SVX_OK; }
    // ---- record discovery ---------------------------------------------------------------------------------------------------------------------------
    // restart points: the first byte of the data (a record start by construction) and the start of every BGZF block behind it
    std::vector<uint64_t> blk;
    blk.push_back(c.data_begin);
    for (size_t k = 0; k < nb_in; k++) { const uint64_t o = DD_HEAD + out_at[k]; if (blocks[k].isize && o > c.data_begin) blk.push_back(o); }
    const long long nb = (long long)blk.size();
    blk.push_back(c.data_end);
    SVXCHK(c.blk_off.reserve((size_t)(nb + 1) * 8)); SVXCHK(c.anchor.reserve((size_t)nb * 8)); SVXCHK(c.cnt.reserve((size_t)nb * 4));
    SVXCHK(c.exit_at.reserve((size_t)nb * 8)); SVXCHK(c.base.reserve((size_t)(nb + 1) * 8));
    SVXCHK(svx_h2d(c.blk_off.p, blk.data(), (size_t)(nb + 1) * 8, st));
    HIPCHK(hipMemsetAsync(d->err.p, 0, 64, st));
    static const bool verify_crc = []() { const char* e = getenv("SVX_BAM_VERIFY_CRC"); return !(e && e[0] == '0'); }();
    if (verify_crc && nb_in) {
        if (!d->crc_shift.p) {
            uint32_t m[CRC_POW][32];
            crc_shift_matrices(m);
            SVXCHK(d->crc_shift.reserve(sizeof m));
            SVXCHK(svx_h2d(d->crc_shift.p, m, sizeof m, st));
        }
        c.crc_host.resize(nb_in);
        for (size_t k = 0; k < nb_in; k++) c.crc_host[k] = CrcJob{(unsigned long long)(DD_HEAD + out_at[k]), blocks[k].isize, blocks[k].crc};
        SVXCHK(c.crc_jobs.reserve(nb_in * sizeof(CrcJob)));
        SVXCHK(svx_h2d(c.crc_jobs.p, c.crc_host.data(), nb_in * sizeof(CrcJob), st));
        k_crc32<<<(unsigned)nb_in, 64, 0, st>>>(sp, c.crc_jobs.as<CrcJob>(), (long long)nb_in, d->crc_shift.as<uint32_t>(), d->err.as<int>() + 8);
    }
    k_anchor<<<(unsigned)nb, 64, 0, st>>>(sp, c.blk_off.as<uint64_t>(), nb, c.data_end, d->n_ref, d->ref_len.as<int32_t>(), c.anchor.as<uint64_t>());
    k_walk<<<GRIDB(nb, 64), 64, 0, st>>>(sp, c.blk_off.as<uint64_t>(), nb, c.data_end, c.anchor.as<uint64_t>(), c.cnt.as<uint32_t>(), c.exit_at.as<uint64_t>(), nullptr, nullptr);
    HIPCHK(hipGetLastError());
    std::vector<uint64_t> anchor((size_t)nb), exit_at((size_t)nb), base((size_t)nb + 1, 0);
    std::vector<uint32_t> cnt((size_t)nb);
    int crc_err[2] = {0, 0};
    {
        HostCopy hc(st);
        SVXCHK(hc.d2h(anchor.data(), c.anchor.p, (size_t)nb * 8)); SVXCHK(hc.d2h(exit_at.data(), c.exit_at.p, (size_t)nb * 8));
        SVXCHK(hc.d2h(cnt.data(), c.cnt.p, (size_t)nb * 4)); SVXCHK(hc.d2h(crc_err, d->err.as<int>() + 8, 8));
        SVXCHK(hc.finish());
    }
    if (crc_err[0]) { char msg[96]; snprintf(msg, sizeof msg, "BGZF block %d of the chunk fails its CRC32", crc_err[1]); return svx_fail(SVX_E_ARG, msg, __FILE__, __LINE__, hipSuccess); }
    // the anchors are right iff they link up: the walk that leaves a block must arrive exactly at the next anchor, and the first anchor is the known start
    bool linked = anchor[0] == c.data_begin;
    uint64_t n_rec = 0, tail = c.data_begin;
    {
        long long prev_b = -1;
        for (long long b = 0; b < nb && linked; b++) {
            base[(size_t)b] = n_rec;
            if (anchor[(size_t)b] == DD_NONE) continue;
            if (prev_b >= 0 && exit_at[(size_t)prev_b] != anchor[(size_t)b]) linked = false;
            n_rec += cnt[(size_t)b];
            prev_b = b;
        }
        if (linked && prev_b >= 0) tail = exit_at[(size_t)prev_b];
        base[(size_t)nb] = n_rec;
    }
    if (linked && tail + 4 <= c.data_end) {
        // the last walk left its block at `tail` and no block behind it produced an anchor: that is only right if the record at `tail` is the
        // incomplete one the chunk ends in (a complete record there means a record start the anchor search did not recognise)
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_TAIL_BS], sp + tail, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const uint32_t bs = (uint32_t)d->h_cnt[DD_H_TAIL_BS];
        if (tail + 4ull + bs <= c.data_end) linked = false;
    }
    if (linked) {
        SVXCHK(c.rec_off.reserve((size_t)(n_rec + 1) * 8));
        SVXCHK(svx_h2d(c.base.p, base.data(), (size_t)(nb + 1) * 8, st));
        k_walk<<<GRIDB(nb, 64), 64, 0, st>>>(sp, c.blk_off.as<uint64_t>(), nb, c.data_end, c.anchor.as<uint64_t>(), nullptr, nullptr, c.base.as<uint64_t>(), c.rec_off.as<uint64_t>());
    } else {
        d->stats.fallbacks++;
        unsigned long long* cn = d->counters.as<unsigned long long>();
        k_walk_serial<<<1, 1, 0, st>>>(sp, c.data_begin, c.data_end, 0, nullptr, cn, reinterpret_cast<uint64_t*>(cn + 1));
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_OPS], cn, 16, hipMemcpyDeviceToHost, st));      // (count, tail: slots DD_H_OPS and DD_H_WALK_TAIL, free at this point of the load)
        HIPCHK(hipStreamSynchronize(st));
        n_rec = d->h_cnt[DD_H_OPS]; tail = d->h_cnt[DD_H_WALK_TAIL];
        SVXCHK(c.rec_off.reserve((size_t)(n_rec + 1) * 8));
        k_walk_serial<<<1, 1, 0, st>>>(sp, c.data_begin, c.data_end, n_rec, c.rec_off.as<uint64_t>(), cn, reinterpret_cast<uint64_t*>(cn + 1));
    }
    HIPCHK(hipGetLastError());
    c.tail_start = (size_t)tail;
    if (final_chunk && tail != c.data_end) return svx_fail(SVX_E_ARG, "truncated BAM record at the end of the file", __FILE__, __LINE__, hipSuccess);
    c.n_rec = (int64_t)n_rec;
    d->stats.t_discover += dd_now() - t0; t0 = dd_now();
    if (n_rec == 0) { HIPCHK(hipStreamSynchronize(st)); c.loaded = true; return SVX_OK; }
    // ---- decode ---------------------------------------------------------------------------------------------------------------------------------------
    const long long n = (long long)n_rec;
    const size_t N1 = (size_t)n + 1;
    SVXCHK(c.desc.reserve((size_t)n * sizeof(RecDesc))); SVXCHK(c.n_cig.reserve(N1 * 4)); SVXCHK(c.n_seg.reserve(N1 * 4)); SVXCHK(c.n_segop.reserve(N1 * 4));
    HIPCHK(hipMemsetAsync(c.n_cig.as<uint32_t>() + n, 0, 4, st)); HIPCHK(hipMemsetAsync(c.n_seg.as<uint32_t>() + n, 0, 4, st)); HIPCHK(hipMemsetAsync(c.n_segop.as<uint32_t>() + n, 0, 4, st));
    k_measure<<<GRIDB(n, 128), 128, 0, st>>>(sp, c.rec_off.as<uint64_t>(), n, min_mapq, c.desc.as<RecDesc>(), c.n_cig.as<uint32_t>(), c.n_seg.as<uint32_t>(), c.n_segop.as<uint32_t>(),
                                            d->err.as<int>());
    HIPCHK(hipGetLastError());
    SVXCHK(c.cigar_off.reserve(N1 * 8)); SVXCHK(c.seg_off.reserve(N1 * 4)); SVXCHK(c.segop_off.reserve(N1 * 8)); SVXCHK(c.name_at.reserve(N1 * 8));
    // 32-bit counts -> 64-bit offsets: widen (into name_at as scratch), then scan
    k_widen_u32<<<GRIDB(n + 1, 256), 256, 0, st>>>(n + 1, c.n_cig.as<uint32_t>(), c.name_at.as<uint64_t>());
    SVXCHK(dd_scan<uint64_t>(d, c, c.name_at.as<uint64_t>(), c.cigar_off.as<uint64_t>(), N1));
    SVXCHK(dd_scan<uint32_t>(d, c, c.n_seg.as<uint32_t>(), c.seg_off.as<uint32_t>(), N1));
    k_widen_u32<<<GRIDB(n + 1, 256), 256, 0, st>>>(n + 1, c.n_segop.as<uint32_t>(), c.name_at.as<uint64_t>());
    SVXCHK(dd_scan<uint64_t>(d, c, c.name_at.as<uint64_t>(), c.segop_off.as<uint64_t>(), N1));
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_OPS], c.cigar_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_SEGOPS], c.segop_off.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_SEG], c.seg_off.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    SVXCHK(dd_check(d, "records"));
    c.tot_ops = (int64_t)d->h_cnt[DD_H_OPS]; c.tot_segops = (int64_t)d->h_cnt[DD_H_SEGOPS]; c.tot_seg = (int64_t)(uint32_t)d->h_cnt[DD_H_SEG];
    SVXCHK(c.flag.reserve(N1 * 2)); SVXCHK(c.tid.reserve(N1 * 4)); SVXCHK(c.pos.reserve(N1 * 4)); SVXCHK(c.mapq.reserve(N1)); SVXCHK(c.lseq.reserve(N1 * 4));
    SVXCHK(c.read_id.reserve(N1 * 4)); SVXCHK(c.seq_off.reserve(N1 * 8)); SVXCHK(c.cigar.reserve((size_t)(c.tot_ops + 16) * 4));
    const size_t S1 = (size_t)c.tot_seg + 1;
    SVXCHK(c.seg_tid.reserve(S1 * 4)); SVXCHK(c.seg_pos.reserve(S1 * 4)); SVXCHK(c.seg_rev.reserve(S1)); SVXCHK(c.seg_mapq.reserve(S1)); SVXCHK(c.seg_lseq.reserve(S1 * 4));
    SVXCHK(c.seg_cigar_off.reserve(S1 * 8)); SVXCHK(c.seg_cigar.reserve((size_t)(c.tot_segops + 16) * 4));
    HIPCHK(hipMemsetAsync(c.seg_cigar_off.p, 0, 8, st));
    HIPCHK(hipMemsetAsync(d->counters.p, 0, 256, st));
    ContigTable ct{d->ct_key.as<uint64_t>(), d->ct_tid.as<int32_t>(), d->ct_mask, d->ct_names.as<char>(), d->ct_name_off.as<uint32_t>()};
    k_fields<<<GRIDB(n, 128), 128, 0, st>>>(sp, c.rec_off.as<uint64_t>(), n, c.desc.as<RecDesc>(), c.seg_off.as<uint32_t>(), c.segop_off.as<uint64_t>(), ct, c.flag.as<uint16_t>(),
                                           c.tid.as<int32_t>(), c.pos.as<int32_t>(), c.mapq.as<uint8_t>(), c.lseq.as<int32_t>(), c.seq_off.as<uint64_t>(), c.seg_tid.as<int32_t>(),
                                           c.seg_pos.as<int32_t>(), c.seg_rev.as<uint8_t>(), c.seg_mapq.as<uint8_t>(), c.seg_lseq.as<int32_t>(), c.seg_cigar_off.as<uint64_t>(),
                                           c.seg_cigar.as<uint32_t>(), d->err.as<int>(), d->counters.as<unsigned long long>() + 4);
    c.seq_end_host = c.data_end;
    SVXCHK(svx_h2d(c.seq_off.as<uint64_t>() + n, &c.seq_end_host, 8, st));
    k_cigar_copy<<<GRIDB(n, 4), 256, 0, st>>>(sp, n, c.desc.as<RecDesc>(), c.cigar_off.as<uint64_t>(), c.cigar.as<uint32_t>());
    HIPCHK(hipGetLastError());
    d->stats.t_decode += dd_now() - t0; t0 = dd_now();
    // ---- read names ---------------------------------------------------------------------------------------------------------------------------------
    {
        const size_t have = d->names.size();
        if ((have + (size_t)n) * 2 > d->nt_cap) {                       // keep the load factor below one half: grow and re-insert
            uint32_t cap = d->nt_cap; while ((have + (size_t)n) * 2 > cap) cap <<= 1;
            DevBuf ok = d->nt_key, oc = d->nt_check, oi = d->nt_id; const uint32_t old_cap = d->nt_cap;
            d->nt_key = DevBuf(); d->nt_check = DevBuf(); d->nt_id = DevBuf();
            SVXCHK(dd_alloc_names(d, cap));
            NameTable nt{d->nt_key.as<unsigned long long>(), d->nt_check.as<unsigned long long>(), d->nt_id.as<int32_t>(), cap - 1};
            k_name_rehash<<<GRIDB(old_cap, 256), 256, 0, st>>>(old_cap, ok.as<unsigned long long>(), oc.as<unsigned long long>(), oi.as<int32_t>(), nt);
            HIPCHK(hipStreamSynchronize(st));
            ok.release(); oc.release(); oi.release();
        }
        NameTable nt{d->nt_key.as<unsigned long long>(), d->nt_check.as<unsigned long long>(), d->nt_id.as<int32_t>(), d->nt_cap - 1};
        SVXCHK(c.slot_of.reserve((size_t)n * 4)); SVXCHK(c.new_rec.reserve((size_t)n * 4)); SVXCHK(c.name_len.reserve(N1 * 4));
        unsigned int* n_new_dev = reinterpret_cast<unsigned int*>(d->counters.as<unsigned long long>() + 8);
        k_name_insert<<<GRIDB(n, 256), 256, 0, st>>>(n, c.desc.as<RecDesc>(), nt, (int32_t)have, c.slot_of.as<uint32_t>(), n_new_dev, c.new_rec.as<uint32_t>());
        k_name_ids<<<GRIDB(n, 256), 256, 0, st>>>(n, c.desc.as<RecDesc>(), nt, c.slot_of.as<uint32_t>(), c.read_id.as<int32_t>(), d->err.as<int>());
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_NEW], n_new_dev, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_SA_BAD], d->counters.as<unsigned long long>() + 4, 8, hipMemcpyDeviceToHost, st));
        SVXCHK(dd_check(d, "fields / names"));
        const long long n_new = (long long)(unsigned int)d->h_cnt[DD_H_NEW];
        if (d->h_cnt[DD_H_SA_BAD]) fprintf(stderr, "WARNING: %llu SA tag entries do not consist of 6 fields. This could be a sign of invalid characters (e.g. commas or semicolons) in a "
                                          "chromosome name of the reference genome.\n", (unsigned long long)d->h_cnt[DD_H_SA_BAD]);
        if (n_new) {
            HIPCHK(hipMemsetAsync(c.name_len.as<uint32_t>() + n_new, 0, 4, st));
            k_name_lens<<<GRIDB(n_new, 256), 256, 0, st>>>(n_new, c.new_rec.as<uint32_t>(), c.desc.as<RecDesc>(), c.name_len.as<uint32_t>());
            k_widen_u32<<<GRIDB(n_new + 1, 256), 256, 0, st>>>(n_new + 1, c.name_len.as<uint32_t>(), c.segop_off.as<uint64_t>());      // (segop_off is free again: scratch)
            SVXCHK(dd_scan<uint64_t>(d, c, c.segop_off.as<uint64_t>(), c.name_at.as<uint64_t>(), (size_t)n_new + 1));
            HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_BLOB], c.name_at.as<uint64_t>() + n_new, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const size_t blob_bytes = (size_t)d->h_cnt[DD_H_BLOB];
            SVXCHK(c.name_blob.reserve(blob_bytes + 16));
            k_name_copy<<<GRIDB(n_new, 256), 256, 0, st>>>(sp, c.rec_off.as<uint64_t>(), n_new, c.new_rec.as<uint32_t>(), c.desc.as<RecDesc>(), c.name_at.as<uint64_t>(), c.name_blob.as<char>());
            std::vector<char> blob(blob_bytes + 1);
            SVXCHK(svx_d2h(blob.data(), c.name_blob.p, blob_bytes, st));
            d->names.reserve(have + (size_t)n_new);
            const char* q = blob.data();
            for (long long k = 0; k < n_new; k++) { const size_t ln = strlen(q); d->names.emplace_back(q, ln); q += ln + 1; }
        }
    }
    d->stats.t_names += dd_now() - t0;
    if (mode == 1 && !final_chunk) {
        // query-name mode: the chunk may end in the middle of a read's group - the last group goes to the next chunk whole (carried over like a partial record)
        unsigned long long* lg = d->counters.as<unsigned long long>() + 16;
        HIPCHK(hipMemsetAsync(lg, 0, 8, st));
        k_q_last_group<<<GRIDB(n, 256), 256, 0, st>>>(n, c.read_id.as<int32_t>(), lg);
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_LAST_GROUP], lg, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const long long L = (long long)d->h_cnt[DD_H_LAST_GROUP];
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_LAST_OFF], c.rec_off.as<uint64_t>() + L, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        c.tail_start = (size_t)d->h_cnt[DD_H_LAST_OFF];
        c.n_rec = L;
    }
    d->stats.records += c.n_rec;
    c.loaded = true;
    return SVX_OK;
}

int devdec_count(svx_devdec* d, int slot, int32_t tid_limit, int64_t* n_rec, int64_t* n_valid) {
    DevChunk& c = d->chunk[slot];
    *n_rec = c.n_rec; *n_valid = c.n_rec;
    if (tid_limit == -2 || c.n_rec == 0) return SVX_OK;
    HIPCHK(hipSetDevice(d->device));
    unsigned long long* first = d->counters.as<unsigned long long>() + 12;
    d->h_cnt[DD_H_FIRST_BEYOND] = (unsigned long long)c.n_rec;            // (pinned slot: the source of an asynchronous copy, not a stack local)
    HIPCHK(hipMemcpyAsync(first, &d->h_cnt[DD_H_FIRST_BEYOND], 8, hipMemcpyHostToDevice, d->stream));
    k_first_beyond<<<GRIDB(c.n_rec, 256), 256, 0, d->stream>>>(c.n_rec, c.tid.as<int32_t>(), tid_limit, first);
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_FIRST_BEYOND], first, 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    *n_valid = (int64_t)d->h_cnt[DD_H_FIRST_BEYOND];
    return SVX_OK;
}

int devdec_batch(svx_devdec* d, int slot, int64_t first, int64_t* count_io, int mode, int min_mapq, svx_batch* out) {
    DevChunk& c = d->chunk[slot];
    int64_t count = *count_io;
    if (!c.loaded || first < 0 || count < 0 || first + count > c.n_rec) return svx_fail(SVX_E_ARG, "record range outside the chunk", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(d->device));
    hipStream_t st = d->batch_stream;
    if (mode == 1 && count > 0 && first + count < c.n_rec) {
        // a read's group is never split: the batch grows to the next group boundary (the chunk itself ends on one)
        // (a counter of its own: devdec_load - which may be running on the loader thread for the NEXT chunk - clears all of d->counters)
        SVXCHK(d->batch_cnt.reserve(64));
        unsigned long long* nb = d->batch_cnt.as<unsigned long long>();
        const unsigned long long big = (unsigned long long)c.n_rec;
        d->h_cnt[DD_H_NEXT_BOUNDARY] = big;
        HIPCHK(hipMemcpyAsync(nb, &d->h_cnt[DD_H_NEXT_BOUNDARY], 8, hipMemcpyHostToDevice, st));
        k_q_next_boundary<<<GRIDB(c.n_rec - (first + count), 256), 256, 0, st>>>(first + count, c.n_rec, c.read_id.as<int32_t>(), nb);
        HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_NEXT_BOUNDARY], nb, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        count = (int64_t)d->h_cnt[DD_H_NEXT_BOUNDARY] - first;
        *count_io = count;
    }
    const int f = c.order_flip; c.order_flip ^= 1;
    SVXCHK(c.order[f].reserve((size_t)(count + 1) * 4)); SVXCHK(c.seg_order[f].reserve((size_t)(count + 1) * 4));
    memset(out, 0, sizeof *out);
    out->on_device = 1; out->n_rec = count;
    out->tid = c.tid.as<int32_t>() + first; out->pos = c.pos.as<int32_t>() + first; out->mapq = c.mapq.as<uint8_t>() + first;
    out->lseq = c.lseq.as<int32_t>() + first; out->read_id = c.read_id.as<int32_t>() + first; out->order = c.order[f].as<uint32_t>(); out->seg_order = c.seg_order[f].as<uint32_t>();
    out->cigar_off = c.cigar_off.as<uint64_t>() + first; out->cigar = c.cigar.as<uint32_t>(); out->seq_off = c.seq_off.as<uint64_t>() + first; out->seq = c.stream.as<uint8_t>();
    out->n_contig = d->n_ref; out->contig_rank = d->contig_rank.as<int32_t>();
    if (mode != 1) {
        if (count) k_order_iota<<<GRIDB(count, 256), 256, 0, st>>>(count, c.order[f].as<uint32_t>(), c.seg_order[f].as<uint32_t>());
        HIPCHK(hipStreamSynchronize(st));
        out->flag = c.flag.as<uint16_t>() + first;
        out->seg_off = c.seg_off.as<uint32_t>() + first; out->n_seg = c.tot_seg; out->seg_tid = c.seg_tid.as<int32_t>(); out->seg_pos = c.seg_pos.as<int32_t>();
        out->seg_rev = c.seg_rev.as<uint8_t>(); out->seg_mapq = c.seg_mapq.as<uint8_t>(); out->seg_lseq = c.seg_lseq.as<int32_t>(); out->seg_cigar_off = c.seg_cigar_off.as<uint64_t>();
        out->seg_cigar = c.seg_cigar.as<uint32_t>();
        return SVX_OK;
    }
    // ---- query-name mode: groups, verdicts, emission slots and the segment table of this batch (kernels k_q_*) ------------------------------------------------
    const long long n = count;
    const size_t N1 = (size_t)n + 1;
    const uint16_t* fl = c.flag.as<uint16_t>() + first; const uint8_t* mq = c.mapq.as<uint8_t>() + first; const int32_t* rid = c.read_id.as<int32_t>() + first;
    SVXCHK(c.q_flag[f].reserve(N1 * 2)); SVXCHK(c.q_seg_off[f].reserve(N1 * 4));
    SVXCHK(c.q_head.reserve(N1 * 4)); SVXCHK(c.q_good.reserve(N1 * 4)); SVXCHK(c.q_gidx.reserve(N1 * 4)); SVXCHK(c.q_good_ex.reserve(N1 * 4));
    k_q_marks<<<GRIDB(n + 1, 256), 256, 0, st>>>(n, fl, mq, rid, min_mapq, c.q_head.as<uint32_t>(), c.q_good.as<uint32_t>());
    SVXCHK(dd_scan<uint32_t>(d, c, c.q_head.as<uint32_t>(), c.q_gidx.as<uint32_t>(), N1, st));
    SVXCHK(dd_scan<uint32_t>(d, c, c.q_good.as<uint32_t>(), c.q_good_ex.as<uint32_t>(), N1, st));
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_GROUPS], c.q_gidx.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const long long G = (long long)(uint32_t)d->h_cnt[DD_H_GROUPS];
    const size_t G1 = (size_t)G + 1;
    SVXCHK(c.q_gs.reserve(G1 * 4)); SVXCHK(c.q_nprim.reserve(G1 * 4)); SVXCHK(c.q_pidx.reserve(G1 * 4)); SVXCHK(c.q_slots.reserve(G1 * 4)); SVXCHK(c.q_rows.reserve(G1 * 4));
    SVXCHK(c.q_slot_ex.reserve(G1 * 4)); SVXCHK(c.q_row_ex.reserve(G1 * 4));
    HIPCHK(hipMemsetAsync(c.q_nprim.p, 0, G1 * 4, st)); HIPCHK(hipMemsetAsync(c.q_pidx.p, 0, G1 * 4, st));
    if (n) k_q_groups<<<GRIDB(n, 256), 256, 0, st>>>(n, c.q_head.as<uint32_t>(), c.q_gidx.as<uint32_t>(), fl, c.q_gs.as<uint32_t>(), c.q_nprim.as<uint32_t>(), c.q_pidx.as<uint32_t>());
    k_q_verdict<<<GRIDB(G + 1, 256), 256, 0, st>>>(G, n, c.q_gs.as<uint32_t>(), c.q_nprim.as<uint32_t>(), c.q_pidx.as<uint32_t>(), fl, mq, min_mapq, c.q_good_ex.as<uint32_t>(),
                                                 c.q_slots.as<uint32_t>(), c.q_rows.as<uint32_t>());
    SVXCHK(dd_scan<uint32_t>(d, c, c.q_slots.as<uint32_t>(), c.q_slot_ex.as<uint32_t>(), G1, st));
    SVXCHK(dd_scan<uint32_t>(d, c, c.q_rows.as<uint32_t>(), c.q_row_ex.as<uint32_t>(), G1, st));
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_ROWS], c.q_row_ex.as<uint32_t>() + G, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const long long R = (long long)(uint32_t)d->h_cnt[DD_H_ROWS];
    const size_t R1 = (size_t)R + 1;
    SVXCHK(c.q_seg_tid[f].reserve(R1 * 4)); SVXCHK(c.q_seg_pos[f].reserve(R1 * 4)); SVXCHK(c.q_seg_rev[f].reserve(R1)); SVXCHK(c.q_seg_mapq[f].reserve(R1));
    SVXCHK(c.q_seg_lseq[f].reserve(R1 * 4)); SVXCHK(c.q_seg_cigar_off[f].reserve(R1 * 8)); SVXCHK(c.q_row_rec.reserve(R1 * 4)); SVXCHK(c.q_row_ops.reserve(R1 * 4));
    HIPCHK(hipMemsetAsync(c.q_row_ops.as<uint32_t>() + R, 0, 4, st));
    k_q_records<<<GRIDB(n + 1, 256), 256, 0, st>>>(n, G, c.q_head.as<uint32_t>(), c.q_gidx.as<uint32_t>(), c.q_gs.as<uint32_t>(), c.q_pidx.as<uint32_t>(), c.q_good.as<uint32_t>(),
                                                 c.q_good_ex.as<uint32_t>(), c.q_slots.as<uint32_t>(), c.q_slot_ex.as<uint32_t>(), c.q_row_ex.as<uint32_t>(), fl, out->tid, out->pos, mq,
                                                 out->lseq, out->cigar_off, c.q_flag[f].as<uint16_t>(), c.order[f].as<uint32_t>(), c.seg_order[f].as<uint32_t>(),
                                                 c.q_seg_off[f].as<uint32_t>(), c.q_seg_tid[f].as<int32_t>(), c.q_seg_pos[f].as<int32_t>(), c.q_seg_rev[f].as<uint8_t>(),
                                                 c.q_seg_mapq[f].as<uint8_t>(), c.q_seg_lseq[f].as<int32_t>(), c.q_row_rec.as<uint32_t>(), c.q_row_ops.as<uint32_t>());
    SVXCHK((svx_exclusive_scan<uint32_t, uint64_t>(c.q_row_ops.as<uint32_t>(), c.q_seg_cigar_off[f].as<uint64_t>(), (long long)R1, st, c.scan_tmp)));
    HIPCHK(hipMemcpyAsync(&d->h_cnt[DD_H_ROW_OPS], c.q_seg_cigar_off[f].as<uint64_t>() + R, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const size_t ops = (size_t)d->h_cnt[DD_H_ROW_OPS];
    SVXCHK(c.q_seg_cigar[f].reserve((ops + 16) * 4));
    if (R) k_q_seg_cigar<<<GRIDB(R, 4), 256, 0, st>>>(R, c.q_row_rec.as<uint32_t>(), out->cigar_off, out->cigar, c.q_seg_cigar_off[f].as<uint64_t>(), c.q_seg_cigar[f].as<uint32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    out->flag = c.q_flag[f].as<uint16_t>();
    out->seg_off = c.q_seg_off[f].as<uint32_t>(); out->n_seg = R; out->seg_tid = c.q_seg_tid[f].as<int32_t>(); out->seg_pos = c.q_seg_pos[f].as<int32_t>();
    out->seg_rev = c.q_seg_rev[f].as<uint8_t>(); out->seg_mapq = c.q_seg_mapq[f].as<uint8_t>(); out->seg_lseq = c.q_seg_lseq[f].as<int32_t>();
    out->seg_cigar_off = c.q_seg_cigar_off[f].as<uint64_t>(); out->seg_cigar = c.q_seg_cigar[f].as<uint32_t>();
    return SVX_OK;
}
