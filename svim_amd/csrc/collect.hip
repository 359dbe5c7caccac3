// collect.hip - COLLECT on the GPU: CIGAR scan (signature emission + alignment geometry), split-read
// segment analysis, ordering and inserted-sequence gather.
//
// Reference being replaced (eldariont/svim v2.0.0):
//   analyze_alignment_file_coordsorted / _querysorted   src/svim/SVIM_COLLECT.py:96-167
//   analyze_cigar_indel / analyze_alignment_indel        src/svim/SVIM_intra.py:8-51
//   analyze_read_segments, is_similar                    src/svim/SVIM_inter.py:11-302
//   pysam accessors reference_end / query_alignment_start,end / infer_read_length (htslib rules)
//
// Kernel 1 (k_cigar_scan) is the HBM-bound kernel of the path: one 64-lane wavefront per alignment streams
// the packed CIGAR words with 16-byte loads per lane (1 KiB per wave-instruction, fully coalesced), keeps the
// running (ref, read) cursors as lane-local partial sums and only falls into the cross-lane prefix scan for
// the rare chunks that actually emit a signature.  No LDS, no MFMA (integer / indexing workload).
#include "common.hpp"
#include <cstdlib>

enum { CNT_SIG = 0, CNT_BND = 1, CNT_USED = 2, CNT_OPS = 3, CNT_SEGOPS = 4, CNT_INS_BASES = 5, CNT_RAW = 6, CNT_OVERFLOW = 7, CNT_SEQ_MISSING = 8 };
#define RAW_SHARDS 8192      /* one private raw-output region per persistent wave: no allocation atomics at all in the scan
                                (a single global counter serialised the launch at ~12 ns per same-address atomic) */

#define KEY(slot, phase, ord) (((uint64_t)(slot) << 32) | ((uint64_t)(phase) << 30) | (uint64_t)(ord))

struct EmitTarget {
    SigPtrs p;
    int64_t cap;
    unsigned long long* counter;
};

__device__ __forceinline__ void write_sig(const EmitTarget& t, long long i, uint64_t key, int type, int src, int aux, int contig,
                                          long long start, long long end, int contig2, long long pos2, int read_id, int rec,
                                          int qpos, int qlen) {
    if (i >= t.cap) return;           // counted but not stored: host grows the buffers and reruns
    t.p.key[i] = key; t.p.type[i] = (uint8_t)type; t.p.src[i] = (uint8_t)src; t.p.aux[i] = (uint8_t)aux;
    t.p.contig[i] = contig; t.p.start[i] = (int)start; t.p.end[i] = (int)end; t.p.contig2[i] = contig2;
    t.p.pos2[i] = (int)pos2; t.p.read_id[i] = read_id; t.p.rec[i] = rec; t.p.qpos[i] = qpos; t.p.qlen[i] = qlen;
}

// What the scan kernel emits for an indel: the heavy signature-table write happens in k_emit_indels, so that the
// streaming loop keeps a small register footprint.
struct RawIndel { uint32_t item; uint32_t opidx; int pos_ref; int pos_read; int len_op; };     // len_op = len<<1 | is_del
struct RawTarget { RawIndel* raw; long long shard_cap; unsigned long long* shard_counter; };     // RAW_SHARDS regions of shard_cap records

// Python slice semantics seq[a:b] on a sequence of length len
__device__ __forceinline__ void py_slice(long long a, long long b, long long len, int& lo, int& n) {
    if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
    if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
    if (b < a) b = a;
    lo = (int)a; n = (int)(b - a);
}

// ------------------------------------------------------------------------------------------------------
// Kernel 1: CIGAR scan.  Work item w < n_rec: BAM record w; w >= n_rec: segment-table row w - n_rec.
// geometry record = {ref_len, query_alignment_start, query_alignment_end, infer_read_length, hard_clipped}
//
// Persistent waves: the grid is sized to the chip (waves = CUs x resident waves), every wave walks items
// w, w + n_waves, ... so the cost of the per-item metadata fetch and of uneven read lengths is spread out.
// Inside an item the wave streams 1 KiB per load instruction (16 B per lane) with the next chunk's load issued
// before the current chunk is decoded (two loads in flight per wave).
// ------------------------------------------------------------------------------------------------------
#ifndef SVX_SCAN_NU
#define SVX_SCAN_NU 2
#endif
#define MASK_REF 0x185      /* M D = X advance the reference cursor (N does not: reference quirk) */
#define MASK_READ 0x193     /* M I S = X advance the read cursor */

__device__ __forceinline__ int op_sel(int mask, int op, int l) { return l & -((mask >> op) & 1); }
// the same on the packed word in ONE instruction per mask: v_bfe_i32 takes its bit offset from the low five bits of a register, i.e. from the operation code and
// the lowest length bit - with the 16-entry table repeated in both halves of the word the length bit does not matter.  -1 for the operations of the table, else 0.
#define TAB32(m) ((int)(((uint32_t)(m) & 0xffffu) | ((uint32_t)(m) << 16)))
__device__ __forceinline__ int op_in(int table32, uint32_t packed) { return __builtin_amdgcn_sbfe(table32, packed, 1u); }

// only what the scan touches: fewer live SGPRs -> more resident blocks per CU (MI355X admits 8 blocks only up to 80 SGPRs)
struct ScanArgs {
    long long n_rec, n_seg;
    const uint16_t* flag; const uint8_t* mapq; const int32_t* lseq; const uint32_t* seg_off; const uint64_t* cigar_off; const uint32_t* cigar;
    const int32_t* seg_lseq; const uint64_t* seg_cigar_off; const uint32_t* seg_cigar;
    int min_mapq, min_sv_size;
};

// One work item as the streaming loop sees it (round 6).  k_scan_prepare writes one 16-byte entry per item; the scan kernel reads an entry with ONE scalar
// load and keeps 32-bit positions relative to the 16-byte-aligned word the item's first operation lies in - the absolute 64-bit offsets of the item, of the
// item after it and of the array ends, and the nine column pointers behind the metadata, were ~45 scalar registers that had to live across the streaming
// loop; with the table the kernel fits the 80 SGPRs / 64 VGPRs of eight waves per SIMD (it is bound by the bytes it keeps in flight).
//   x = low word of a0 = off0 & ~3 (operations);  y = bits 0..23 high bits of a0, bit 24 skip (filtered-out record / short segment row), bit 25 the item
//   needs its geometry record, bits 28..29 lead = off0 & 3;  z = len = off1 - a0;  w = lim = operations of the array from a0 on (clamped): a 16-byte load
//   at position k is legal iff k + 4 <= lim.
#define ITEM_SKIP (1u << 24)
#define ITEM_GEOM (1u << 25)
struct ItemMeta { const uint32_t* base; uint32_t lead, len, lim, flags; };

// Segment rows come from SA tags and most aligners abbreviate their CIGARs to a handful of operations: a wave per row would spend its time on
// latency (one trip, one serial epilogue), so rows of at most SEG_SMALL operations are done one per LANE here and k_cigar_scan skips them.
#define SEG_SMALL 32

// the four operations at position k of an item (k a multiple of 4).  One load instruction whatever the position: where the 16 bytes would run off the end of
// the ARRAY (its last three words) the load is moved back to the array's last four words and the elements are shifted down; what slides in from behind is at a
// position >= len, which the caller masks.  A lane at or behind the item's end does not load (those bytes are the neighbours', fetched by their own waves).
// (lim >= 4: svx_collect_impl pads an array of fewer than four operations.)
__device__ __forceinline__ uint4 load_chunk(const uint32_t* base, uint32_t k, uint32_t len, uint32_t lim) {
    uint4 q = make_uint4(15u, 15u, 15u, 15u);            // op 15 / len 0 = no-op
    if (k < len) {
        const int kk = k + 4 <= lim ? (int)k : (int)lim - 4;            // (signed: the item may start inside the array's last four words)
        q = *reinterpret_cast<const uint4*>(base + kk);
        const uint32_t d = k - (uint32_t)kk;
        if (d) {
            q.x = d == 1 ? q.y : (d == 2 ? q.z : q.w);
            q.y = d == 1 ? q.z : q.w;
            q.z = q.w;
        }
    }
    return q;
}

// geometry record of one alignment from the sums over its CIGAR: {reference length, query_alignment_start, query_alignment_end, infer_read_length,
// hard-clipped bases} with htslib's / pysam's rules; executed by ONE lane
__device__ __forceinline__ void finish_geom(const uint32_t* c, long long n, int lseq, long long sum_ref, long long sum_read, long long sum_n, long long sum_h,
                                            long long sum_s, int* geom_out) {
    long long qstart = 0;
    for (long long i = 0; i < n; i++) {                     // leading clips: hard skipped, soft summed
        const int op = c[i] & 15;
        if (op == 5) continue;
        if (op == 4) qstart += c[i] >> 4; else break;
    }
    long long qend;
    if (lseq == 0) {
        // no stored sequence: M+I+=+X, plus a soft clip met while the running total is still zero
        qend = sum_read - sum_s;
        for (long long i = 0; i < n; i++) {
            const int op = c[i] & 15; const long long l = c[i] >> 4;
            if (l == 0) continue;
            if (op == 4) { qend += l; break; }
            if (op == 0 || op == 1 || op == 7 || op == 8) break;
        }
    } else {
        qend = lseq;
        for (long long i = n - 1; i >= 1; i--) {            // element 0 is never inspected (pysam getQueryEnd)
            const int op = c[i] & 15;
            if (op == 5) continue;
            if (op == 4) qend -= c[i] >> 4; else break;
        }
    }
    long long ref_len = sum_ref + sum_n;
    if (ref_len == 0) ref_len = 1;                          // bam_endpos never returns pos itself
    geom_out[0] = (int)ref_len; geom_out[1] = (int)qstart; geom_out[2] = (int)qend;
    geom_out[3] = (n > 0) ? (int)(sum_read + sum_h) : 0; geom_out[4] = (int)sum_h;
}

// One thread per item: the table entry of the scan (above); the geometry of a short segment row; and, for an item whose geometry the scan computes, the
// stored sequence length in word 0 of its geometry record, where the scan's epilogue picks it up (finish_geom reads it before it writes the record).
__global__ __launch_bounds__(256) void k_scan_prepare(ScanArgs b, unsigned long long total_ops, unsigned long long total_seg_ops, uint4* items, int* geom,
                                                      unsigned long long* shard_counter) {
    const long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (shard_counter && w < RAW_SHARDS) shard_counter[w] = 0ull;          // the scan's per-wave emission counters (a launch of its own before: one gap less in front of the scan)
    if (w >= b.n_rec + b.n_seg) return;
    const bool is_rec = w < b.n_rec;
    const long long s = w - b.n_rec;
    const unsigned long long off0 = is_rec ? b.cigar_off[w] : b.seg_cigar_off[s], off1 = is_rec ? b.cigar_off[w + 1] : b.seg_cigar_off[s + 1];
    const unsigned long long tot = is_rec ? total_ops : total_seg_ops;
    const int lseq = is_rec ? b.lseq[w] : b.seg_lseq[s];
    bool skip, geom_needed;
    if (is_rec) {
        const unsigned flag = b.flag[w];
        skip = (flag & SVX_FLAG_USED_MASK) || (int)b.mapq[w] < b.min_mapq;
        geom_needed = !(flag & 2048u) && b.seg_off[w + 1] > b.seg_off[w];
    } else {
        skip = off1 - off0 <= SEG_SMALL;
        geom_needed = true;
    }
    const unsigned long long a0 = off0 & ~3ull, room = tot - a0;
    uint4 e;
    e.x = (uint32_t)a0;
    e.y = (uint32_t)(a0 >> 32) | (skip ? ITEM_SKIP : 0u) | (geom_needed ? ITEM_GEOM : 0u) | ((uint32_t)(off0 - a0) << 28);
    e.z = (uint32_t)(off1 - a0);
    e.w = room > 0xffffffffull ? 0xffffffffu : (uint32_t)room;
    items[w] = e;
    if (!is_rec && skip) {
        const uint32_t* c = b.seg_cigar + off0;
        const long long n = (long long)(off1 - off0);
        int sum_ref = 0, sum_read = 0, sum_n = 0, sum_h = 0, sum_s = 0;
        for (long long i = 0; i < n; i++) {
            const int op = (int)(c[i] & 15u), l = (int)(c[i] >> 4);
            sum_ref += op_sel(MASK_REF, op, l);
            sum_read += op_sel(MASK_READ, op, l);
            sum_n += (op == 3) ? l : 0;
            sum_h += (op == 5) ? l : 0;
            sum_s += (op == 4) ? l : 0;
        }
        finish_geom(c, n, lseq, sum_ref, sum_read, sum_n, sum_h, sum_s, geom + 5 * w);
    } else if (!skip && geom_needed) geom[5 * w] = lseq;
}

// One item, streamed.  The two cursors are lane-local running sums kept in EVERY chunk (until round 5 the records without segment rows only asked "is any
// of these operations a long I / D" and re-decoded the chunks since the last report when one turned up: 17 % of the bytes were fetched twice, much of it
// from HBM - 768 waves per XCD stream the 4 MB of its L2 between the first read of a chunk and the walk back).  GEOM adds the three sums the geometry
// record needs and its serial epilogue.
__device__ __forceinline__ void scan_item(const bool GEOM, RawIndel* out_raw, int shard_cap, uint32_t w, const ItemMeta& mt, bool need_indel, int* geom_out,
                                          int min_len, int& n_out, uint4 (&nx)[SVX_SCAN_NU], const ItemMeta& mtn) {
    const int lane = lane_id();
    const uint32_t* base = mt.base;
    const uint32_t lead = mt.lead, len = mt.len, lim = mt.lim;
    // the item after this one (len 0 when there is none): its first trip is requested while this item's last trip is decoded
    const uint32_t* base_n = mtn.base;
    const uint32_t len_n = mtn.len, lim_n = mtn.lim;
    int acc_ref = 0, acc_read = 0, acc_n = 0, acc_h = 0, acc_s = 0;
    const int emit_thr = (min_len > 0 ? (min_len > 0x10000000 ? 0x10000000 : min_len) : 0) + 1;       // lengths are 28 bits: (length + 1) >= emit_thr <=> length >= min_len
    // NU consecutive 1 KiB chunks per trip: the loads of the next trip are all issued before the current one is decoded,
    // so a wave keeps NU KiB in flight (memory-level parallelism is what this kernel lives on)
    constexpr int NU = SVX_SCAN_NU;
    uint32_t kb = 0;
    do {                                               // (an empty CIGAR makes one trip too: nothing to decode, but the pipeline stays primed)
        uint4 cu[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) cu[u] = nx[u];
        {
            // one load site: the next trip of this item, or the first trip of the item after it (scalar selects)
            const bool more = len > kb && len - kb > 256u * NU;
            const uint32_t* lb = more ? base : base_n;
            const uint32_t lk = more ? kb + 256u * NU : 0u, ll = more ? len : len_n, lm = more ? lim : lim_n;
#pragma unroll
            for (int u = 0; u < NU; u++) nx[u] = load_chunk(lb, lk + 256u * u + (uint32_t)lane * 4, ll, lm);
        }
#pragma unroll
        for (int u = 0; u < NU; u++) {
        const uint32_t k0 = kb + 256u * u;
        if (k0 >= len) break;
        const uint32_t k = k0 + (uint32_t)lane * 4;
        uint4 q = cu[u];
        // mask the elements outside [lead, len) (only the first and the last chunk of an item can have any)
        if (k0 == 0 || len - k0 < 256u) {
            if (k < lead || k >= len) q.x = 15u;
            if (k + 1 < lead || k + 1 >= len) q.y = 15u;
            if (k + 2 < lead || k + 2 >= len) q.z = 15u;
            if (k + 3 < lead || k + 3 >= len) q.w = 15u;
        }
        const uint32_t v[4] = {q.x, q.y, q.z, q.w};
        int t_ref = 0, t_read = 0;
        // "is any of these a reportable I / D": the largest (length + 1) over the I / D operations against emit_thr (one compare per chunk)
        int id_max = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int l = (int)(v[j] >> 4);
            t_ref += l & op_in(TAB32(MASK_REF), v[j]);
            t_read += l & op_in(TAB32(MASK_READ), v[j]);
            const int sc = (l + 1) & op_in(TAB32(0x6), v[j]);
            id_max = sc > id_max ? sc : id_max;
        }
        const bool any_emit = id_max >= emit_thr;
        if (GEOM) {                                        // (uniform: one copy of the loop serves both kinds of item)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int l = (int)(v[j] >> 4);
                acc_n += l & op_in(TAB32(1 << 3), v[j]);
                acc_h += l & op_in(TAB32(1 << 5), v[j]);
                acc_s += l & op_in(TAB32(1 << 4), v[j]);
            }
        }
        if (need_indel && __any(any_emit)) {
            // rare path: exact cursor positions = wave-wide sum of everything before this chunk + exclusive scan inside it
            const int base_ref = wave_sum_i32(acc_ref), base_read = wave_sum_i32(acc_read);
            const int ex_ref = wave_incl_scan_i32(t_ref) - t_ref, ex_read = wave_incl_scan_i32(t_read) - t_read;
            int pr = 0, pq = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int op = (int)(v[j] & 15u), l = (int)(v[j] >> 4);
                const bool e = ((unsigned)(op - 1) < 2u) && l >= min_len;
                const unsigned long long m = __ballot(e);
                if (m) {
                    const int slot = n_out + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (e && slot < shard_cap) {
                        RawIndel ri;
                        ri.item = w; ri.opidx = (k + j) - lead;
                        ri.pos_ref = base_ref + ex_ref + pr; ri.pos_read = base_read + ex_read + pq; ri.len_op = (l << 1) | (op == 2);
                        out_raw[slot] = ri;
                    }
                    n_out += __popcll(m);
                }
                pr += op_sel(MASK_REF, op, l); pq += op_sel(MASK_READ, op, l);
            }
        }
        acc_ref += t_ref; acc_read += t_read;
        }
        kb += 256u * NU;
    } while (kb < len);
    if (!GEOM) return;
    const long long sum_ref = wave_sum_i32(acc_ref), sum_read = wave_sum_i32(acc_read);
    const long long sum_n = wave_sum_i32(acc_n), sum_h = wave_sum_i32(acc_h), sum_s = wave_sum_i32(acc_s);
    if (lane == 0) finish_geom(base + lead, len > lead ? (long long)(len - lead) : 0ll, geom_out[0] /* the stored sequence length: k_scan_prepare */, sum_ref, sum_read, sum_n, sum_h,
                               sum_s, geom_out);
}

__device__ __forceinline__ ItemMeta load_meta(const uint4* items, uint32_t w, uint32_t n_rec, const uint32_t* cigar, const uint32_t* seg_cigar) {
    const uint4 e = items[w];                               // uniform address: one scalar load
    // (select on integers: a pointer chosen in the two arms of a branch becomes a two-entry table in scratch memory, indexed through a VGPR - and with it
    // everything derived from the base pointer leaves the scalar registers)
    const unsigned long long cigp = w < n_rec ? (unsigned long long)cigar : (unsigned long long)seg_cigar;
    ItemMeta m;
    m.base = reinterpret_cast<const uint32_t*>(cigp) + (((unsigned long long)(e.y & 0xffffffu) << 32) | e.x);
    m.lead = e.y >> 28; m.len = e.z; m.lim = e.w; m.flags = e.y;
    return m;
}

__device__ __forceinline__ void scan_items(const uint4* items, uint32_t n_rec, uint32_t n_items, const uint32_t* cigar, const uint32_t* seg_cigar, int min_len,
                                           const RawTarget& out, int* geom, uint32_t block, uint32_t n_blocks) {
    // items, waves and raw slots are 32-bit counts (a batch holds fewer than 2^31 records + segment rows: svx_collect_impl checks)
    const uint32_t n_waves = n_blocks * 4;
    // the wave index is uniform: keep it (and everything derived from it: metadata, base pointers) in scalar registers
    const uint32_t wave = block * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int n_out = 0;                                 // records this wave has emitted (uniform); its private raw region is shard `wave` (the grid is capped at RAW_SHARDS waves)
    // wave walks items wave, wave + W, ...; first used item of this wave
    uint32_t w = wave;
    ItemMeta mt;
    for (;; w += n_waves) {
        if (w >= n_items) return;
        mt = load_meta(items, w, n_rec, cigar, seg_cigar);
        if (!(mt.flags & ITEM_SKIP)) break;
    }
    RawIndel* out_raw = out.raw + (long long)wave * out.shard_cap;
    const int shard_cap = (int)(out.shard_cap > 0x7fffffffll ? 0x7fffffffll : out.shard_cap);
    uint4 nx[SVX_SCAN_NU];
#pragma unroll
    for (int u = 0; u < SVX_SCAN_NU; u++) nx[u] = load_chunk(mt.base, 256u * u + (uint32_t)lane_id() * 4, mt.len, mt.lim);
    for (;;) {
        // next used item (its table entry comes through the scalar cache)
        uint32_t wn = w;
        ItemMeta mtn = mt;
        bool has_next = false;
        while (n_items - wn > n_waves) {
            wn += n_waves;
            mtn = load_meta(items, wn, n_rec, cigar, seg_cigar);
            if (!(mtn.flags & ITEM_SKIP)) { has_next = true; break; }
        }
        if (!has_next) mtn.len = 0;
        int* geom_out = geom + 5ll * w;
        scan_item((mt.flags & ITEM_GEOM) != 0u, out_raw, shard_cap, w, mt, w < n_rec, geom_out, min_len, n_out, nx, mtn);
        if (!has_next) break;
        w = wn; mt = mtn;
    }
    if (lane_id() == 0) out.shard_counter[wave] = (unsigned long long)n_out;
}

// built for eight waves per SIMD (<= 64 VGPRs, <= 80 SGPRs on gfx950; SVX_SCAN_WAVES=0: the compiler's own choice)
#ifndef SVX_SCAN_WAVES
#define SVX_SCAN_WAVES 8
#endif
#if SVX_SCAN_WAVES > 0
#define SVX_SCAN_ATTR __attribute__((amdgpu_waves_per_eu(SVX_SCAN_WAVES, SVX_SCAN_WAVES)))
#else
#define SVX_SCAN_ATTR
#endif
__global__ __launch_bounds__(256) SVX_SCAN_ATTR void k_cigar_scan(const uint4* items, uint32_t n_rec, uint32_t n_items, const uint32_t* cigar, const uint32_t* seg_cigar,
                                                                  int min_len, RawTarget out, int* geom) {
    scan_items(items, n_rec, n_items, cigar, seg_cigar, min_len, out, geom, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(1024) void k_shard_prefix(const unsigned long long* shard_counter, long long shard_cap, long long* prefix,
                                                       unsigned long long* counters) {
    __shared__ long long s[1024];
    const int t = threadIdx.x;
    constexpr int PER = RAW_SHARDS / 1024;
    long long loc[PER]; long long sum = 0; unsigned long long over = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        unsigned long long c = shard_counter[t * PER + k];
        if ((long long)c > shard_cap) { over += c - (unsigned long long)shard_cap; c = (unsigned long long)shard_cap; }
        loc[k] = sum; sum += (long long)c;
    }
    if (over) atomicAdd(&counters[CNT_OVERFLOW], over);
    s[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const long long v = (t >= o) ? s[t - o] : 0; __syncthreads(); s[t] += v; __syncthreads(); }
    const long long base = s[t] - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) prefix[t * PER + k] = base + loc[k];
    if (t == 1023) { prefix[RAW_SHARDS] = s[t]; counters[CNT_SIG] = (unsigned long long)s[t]; counters[CNT_RAW] = (unsigned long long)s[t]; }
}

// raw indel i of shard g -> signature table slot prefix[g] + i (analyze_alignment_indel, src/svim/SVIM_intra.py:33-51)
__global__ __launch_bounds__(256) void k_emit_indels(svx_batch b, svx_params p, const RawIndel* raw, const unsigned long long* shard_counter,
                                                     long long shard_cap, const long long* prefix, EmitTarget sig, EmitTarget bnd) {
    const int g = blockIdx.y;
    const long long li = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long cnt = (long long)shard_counter[g];
    if (cnt > shard_cap) cnt = shard_cap;
    if (li >= cnt) return;
    const long long i = prefix[g] + li;
    const RawIndel ri = raw[(long long)g * shard_cap + li];
    const long long r = ri.item;
    const int l = ri.len_op >> 1;
    const long long rs = b.pos[r];
    const uint64_t key = KEY(b.order[r], 0, ri.opidx);
    if (ri.len_op & 1) {
        write_sig(sig, i, key, SVX_DEL, SVX_SRC_CIGAR, 0, b.tid[r], rs + ri.pos_ref, rs + ri.pos_ref + l, -1, 0, b.read_id[r], -1, ri.pos_read, 0);
        if (p.all_bnds) {
            const long long bi = (long long)atomicAdd(bnd.counter, 1ull);
            write_sig(bnd, bi, key, SVX_BND, SVX_SRC_CIGAR, 0, b.tid[r], rs + ri.pos_ref, rs + ri.pos_ref + 1, b.tid[r], rs + ri.pos_ref + l, b.read_id[r],
                      -1, 0, 0);
        }
    } else {
        int qpos, qlen;
        py_slice(ri.pos_read, (long long)ri.pos_read + l, b.lseq[r], qpos, qlen);
        write_sig(sig, i, key, SVX_INS, SVX_SRC_CIGAR, 0, b.tid[r], rs + ri.pos_ref, rs + ri.pos_ref + l, -1, 0, b.read_id[r], (int)r, qpos, qlen);
    }
}

// bookkeeping for the metric (reads passing the filter, their CIGAR ops): one thread per record, one atomic pair per
// block - kept out of the scan kernel, where a per-wave atomic on one address serialises the whole launch
__global__ __launch_bounds__(256) void k_count_used(svx_batch b, svx_params p, unsigned long long* counters) {
    // grid-stride: at most 1024 blocks, i.e. 2048 atomics on the two counters whatever the batch size (one pair per block of a 1.2 M-record batch were 9700
    // same-address atomics of ~12 ns each: 0.12 ms)
    unsigned long long used = 0, ops = 0;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < b.n_rec; r += (long long)gridDim.x * blockDim.x) {
        const unsigned f = b.flag[r];
        if (!(f & SVX_FLAG_USED_MASK) && (int)b.mapq[r] >= p.min_mapq) { used += 1; ops += b.cigar_off[r + 1] - b.cigar_off[r]; }
    }
    used = (unsigned long long)wave_sum_i64((long long)used); ops = (unsigned long long)wave_sum_i64((long long)ops);
    __shared__ unsigned long long su[4], so[4];
    if (lane_id() == 0) { su[threadIdx.x >> 6] = used; so[threadIdx.x >> 6] = ops; }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(&counters[CNT_USED], su[0] + su[1] + su[2] + su[3]); atomicAdd(&counters[CNT_OPS], so[0] + so[1] + so[2] + so[3]); }
}

// ------------------------------------------------------------------------------------------------------
// Kernel 2: split-read analysis, one lane per primary record that owns segment-table rows.
// ------------------------------------------------------------------------------------------------------
struct ASeg { int qs, qe, rs, re, tid, rev; };
struct TDup { int chr, s, e, full, fwd, pad; };
struct Trn { int d1, d2, c1, p1, c2, p2; };

__device__ __forceinline__ bool is_similar_d(int chr1, double s1, double e1, int chr2, double s2, double e2, double thr) {
    const double span1 = e1 - s1, span2 = e2 - s2;
    const double c1 = floor((s1 + e1) / 2.0), c2 = floor((s2 + e2) / 2.0);
    const double pd = fabs(c1 - c2) / 900.0;
    const double mx = span1 > span2 ? span1 : span2;
    const double sd = fabs(span1 - span2) / mx;
    return chr1 == chr2 && pd + sd < thr;
}

__device__ __forceinline__ void push_bnd(const EmitTarget& t, uint64_t key, const int* rank, int c1, long long p1, int rev1, int c2,
                                         long long p2, int rev2, int read_id) {
    const long long i = (long long)atomicAdd(t.counter, 1ull);
    const bool keep = (rank[c1] < rank[c2]) || (c1 == c2 && p1 < p2);
    if (keep) write_sig(t, i, key, SVX_BND, SVX_SRC_SUPPL, (rev1 ? 1 : 0) | (rev2 ? 2 : 0), c1, p1, p1 + 1, c2, p2, read_id, -1, 0, 0);
    else      write_sig(t, i, key, SVX_BND, SVX_SRC_SUPPL, (rev2 ? 0 : 1) | (rev1 ? 0 : 2), c2, p2, p2 + 1, c1, p1, read_id, -1, 0, 0);
}

__global__ __launch_bounds__(256) void k_segments(svx_batch b, svx_params p, EmitTarget sig, EmitTarget bnd, const int* rec_geom,
                                                  const int* seg_geom, ASeg* ws_al, TDup* ws_td, Trn* ws_tr) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_rec) return;
    const unsigned f = b.flag[r];
    if ((f & SVX_FLAG_USED_MASK) || (f & 2048u) || (int)b.mapq[r] < p.min_mapq) return;
    long long s0 = b.seg_off[r], s1 = b.seg_off[r + 1];
    if (s1 <= s0) return;                                       // a single alignment has no adjacent pair
    const int* pg = rec_geom + 5 * r;
    if ((f & SVX_FLAG_SA) && pg[4] > 0) return;                 // hard-clipped primary: SA rebuild void (SVIM_COLLECT.py:47)
    ASeg* al = ws_al + s0 + r; TDup* td = ws_td + s0 + r; Trn* tr = ws_tr + s0 + r;
    int na = 0;
    for (long long k = -1; k < s1 - s0; k++) {
        const int* g; int rev, tid; long long pos;
        if (k < 0) { g = pg; rev = (f & 16u) != 0; tid = b.tid[r]; pos = b.pos[r]; }
        else {
            const long long s = s0 + k;
            if ((int)b.seg_mapq[s] < p.min_mapq) continue;       // good_suppl_alns (SVIM_COLLECT.py:113,154)
            g = seg_geom + 5 * s; rev = b.seg_rev[s]; tid = b.seg_tid[s]; pos = b.seg_pos[s];
        }
        ASeg a;
        if (rev) {
            if (g[3] <= 0) continue;                             // infer_read_length() is None
            a.qs = g[3] - g[2]; a.qe = g[3] - g[1];
        } else { a.qs = g[1]; a.qe = g[2]; }
        a.tid = tid; a.rs = (int)pos; a.re = (int)(pos + g[0]); a.rev = rev;
        // stable insertion by (q_start, q_end)
        int j = na - 1;
        while (j >= 0 && (al[j].qs > a.qs || (al[j].qs == a.qs && al[j].qe > a.qe))) { al[j + 1] = al[j]; j--; }
        al[j + 1] = a; na++;
    }
    int ntd = 0, ntr = 0;
    const int* rank = b.contig_rank;
    const uint32_t slot = b.seg_order[r];
    const int rid = b.read_id[r];
    const long long MIN = p.min_sv_size, MAX = p.max_sv_size, GAP = p.segment_gap_tolerance, OVL = p.segment_overlap_tolerance;
    const long long prim_len = b.lseq[r], prim_infer = pg[3];
#define SIG(TYPE_, AUX_, C_, S_, E_) do { const long long i_ = (long long)atomicAdd(sig.counter, 1ull); \
        write_sig(sig, i_, KEY(slot, 1, idx), TYPE_, SVX_SRC_SUPPL, AUX_, C_, S_, E_, -1, 0, rid, -1, 0, 0); } while (0)
#define BND_MAIN(C1_, P1_, R1_, C2_, P2_, R2_) do { push_bnd(sig, KEY(slot, 1, idx), rank, C1_, P1_, R1_, C2_, P2_, R2_, rid); \
        Trn t_; t_.d1 = R1_; t_.d2 = R2_; t_.c1 = C1_; t_.p1 = (int)(P1_); t_.c2 = C2_; t_.p2 = (int)(P2_); tr[ntr++] = t_; } while (0)
#define BND_SIDE(C1_, P1_, R1_, C2_, P2_, R2_) do { if (p.all_bnds) push_bnd(bnd, KEY(slot, 1, idx), rank, C1_, P1_, R1_, C2_, P2_, R2_, rid); } while (0)
#define TDUP(C_, S_, E_, FU_, FW_) do { TDup t_; t_.chr = C_; t_.s = (int)(S_); t_.e = (int)(E_); t_.full = FU_; t_.fwd = FW_; t_.pad = 0; td[ntd++] = t_; } while (0)
    for (int idx = 0; idx + 1 < na; idx++) {
        const ASeg cu = al[idx], nx = al[idx + 1];
        const long long dr = (long long)nx.qs - cu.qe;
        if (cu.tid == nx.tid) {
            const int chr = cu.tid;
            if (cu.rev == nx.rev) {
                const long long dref = cu.rev ? (long long)cu.rs - nx.re : (long long)nx.rs - cu.re;
                if (dr >= -OVL) {
                    if (dref >= -OVL) {
                        const long long dev = dr - dref;
                        if (dev >= MIN) {                                            // INS candidate (SVIM_inter.py:80-94)
                            if (dref <= GAP) {
                                const long long st = cu.rev ? cu.rs : cu.re;
                                const long long a = cu.rev ? prim_infer - nx.qs : cu.qe;
                                int qpos = 0, qlen = 0;
                                if (prim_len > 0) py_slice(a, a + dev, prim_len, qpos, qlen);
                                const long long i_ = (long long)atomicAdd(sig.counter, 1ull);
                                write_sig(sig, i_, KEY(slot, 1, idx), SVX_INS, SVX_SRC_SUPPL, 0, chr, st, st + dev, -1, 0, rid, (int)r, qpos, qlen);
                            }
                        } else if (-MAX <= dev && dev <= -MIN) {                     // DEL candidate (:96-106)
                            if (dr <= GAP) {
                                const long long st = cu.rev ? nx.re : cu.re;
                                SIG(SVX_DEL, 0, chr, st, st - dev);
                                BND_SIDE(chr, st - 1, 0, chr, st - dev, 0);
                            }
                        } else if (dev < -MAX) {                                     // very large DEL or TRANS (:108-116)
                            if (dr <= GAP) {
                                if (!cu.rev) BND_MAIN(chr, (long long)cu.re - 1, 0, chr, (long long)nx.rs, 0);
                                else         BND_MAIN(chr, (long long)cu.rs, 1, chr, (long long)nx.re - 1, 1);
                            }
                        }
                    } else if (dref <= -MIN) {                                       // overlap on the reference (:118-150)
                        if (!cu.rev) {
                            if (nx.re > cu.rs) { TDUP(chr, nx.rs, cu.re, 1, 1); BND_SIDE(chr, (long long)cu.re - 1, 0, chr, (long long)nx.rs, 0); }
                            else if (dref >= -MAX) { TDUP(chr, nx.rs, cu.re, 0, 1); BND_SIDE(chr, (long long)cu.re - 1, 0, chr, (long long)nx.rs, 0); }
                            else BND_MAIN(chr, (long long)cu.re - 1, 0, chr, (long long)nx.rs, 0);
                        } else {
                            if (nx.rs < cu.re) { TDUP(chr, cu.rs, nx.re, 1, 0); BND_SIDE(chr, (long long)cu.rs, 1, chr, (long long)nx.re - 1, 1); }
                            else if (dref >= -MAX) { TDUP(chr, cu.rs, nx.re, 0, 0); BND_SIDE(chr, (long long)cu.rs, 1, chr, (long long)nx.re - 1, 1); }
                            else BND_MAIN(chr, (long long)cu.rs, 1, chr, (long long)nx.re - 1, 1);
                        }
                    }
                }
            } else if (!cu.rev && nx.rev) {                                          // normal -> reverse (:154-178)
                if (-OVL <= dr && dr <= GAP) {
                    if ((long long)nx.rs - cu.re >= -OVL) {
                        const long long sz = (long long)nx.re - cu.re;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_LEFT_FWD, chr, cu.re, nx.re); BND_SIDE(chr, (long long)cu.re - 1, 0, chr, (long long)nx.re - 1, 1); }
                        else if (sz > MAX) BND_MAIN(chr, (long long)cu.re - 1, 0, chr, (long long)nx.re - 1, 1);
                    } else if ((long long)cu.rs - nx.re >= -OVL) {
                        const long long sz = (long long)cu.re - nx.re;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_LEFT_REV, chr, nx.re, cu.re); BND_SIDE(chr, (long long)cu.re - 1, 0, chr, (long long)nx.re - 1, 1); }
                        else if (sz > MAX) BND_MAIN(chr, (long long)cu.re - 1, 0, chr, (long long)nx.re - 1, 1);
                    }
                }
            } else {                                                                 // reverse -> normal (:180-204)
                if (-OVL <= dr && dr <= GAP) {
                    if ((long long)nx.rs - cu.re >= -OVL) {
                        const long long sz = (long long)nx.rs - cu.rs;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_RIGHT_FWD, chr, cu.rs, nx.rs); BND_SIDE(chr, (long long)cu.rs, 1, chr, (long long)nx.rs, 0); }
                        else if (sz > MAX) BND_MAIN(chr, (long long)cu.rs, 1, chr, (long long)nx.rs, 0);
                    } else if ((long long)cu.rs - nx.re >= -OVL) {
                        const long long sz = (long long)cu.rs - nx.rs;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_RIGHT_REV, chr, nx.rs, cu.rs); BND_SIDE(chr, (long long)cu.rs, 1, chr, (long long)nx.rs, 0); }
                        else if (sz > MAX) BND_MAIN(chr, (long long)cu.rs, 1, chr, (long long)nx.rs, 0);
                    }
                }
            }
        } else if (dr >= -OVL && dr <= GAP) {                                        // different chromosomes (:206-240)
            if (cu.rev == nx.rev) {
                if (!cu.rev) BND_MAIN(cu.tid, (long long)cu.re - 1, 0, nx.tid, (long long)nx.rs, 0);
                else         BND_MAIN(cu.tid, (long long)cu.rs, 1, nx.tid, (long long)nx.re - 1, 1);
            } else {
                if (!cu.rev) BND_MAIN(cu.tid, (long long)cu.re - 1, 0, nx.tid, (long long)nx.re - 1, 1);
                else         BND_MAIN(cu.tid, (long long)cu.rs, 1, nx.tid, (long long)nx.rs, 0);
            }
        }
    }
#undef SIG
#undef BND_MAIN
#undef BND_SIDE
#undef TDUP
    // tandem-duplication runs (:242-272); current_direction is deliberately never refreshed (reference quirk)
    if (ntd > 0) {
        int cur_chr = td[0].chr; long long sum_s = td[0].s, sum_e = td[0].e, cnt = 1; int any_full = td[0].full;
        const int cur_dir = td[0].fwd; unsigned ord = 0;
        for (int k = 1; k <= ntd; k++) {
            bool merge = false;
            if (k < ntd) {
                const double ms = (double)sum_s / (double)cnt, me = (double)sum_e / (double)cnt;
                merge = is_similar_d(cur_chr, ms, me, td[k].chr, (double)td[k].s, (double)td[k].e, 0.3) && cur_dir == td[k].fwd;
            }
            if (merge) { sum_s += td[k].s; sum_e += td[k].e; cnt++; any_full |= td[k].full; }
            else {
                const long long i_ = (long long)atomicAdd(sig.counter, 1ull);
                write_sig(sig, i_, KEY(slot, 2, ord), SVX_DUP_TAN, SVX_SRC_SUPPL, any_full ? 1 : 0, cur_chr, sum_s / cnt, sum_e / cnt, -1, cnt,
                          rid, -1, 0, 0);
                ord++;
                if (k < ntd) { cur_chr = td[k].chr; sum_s = td[k].s; sum_e = td[k].e; cnt = 1; any_full = td[k].full; }
            }
        }
    }
    // insertions with detected origin (:274-300)
    for (int ti = 0; ti < ntr; ti++) {
        const Trn t = tr[ti];
        for (int bi = 0; bi < ti; bi++) {
            const Trn q = tr[bi];
            if (q.d1 == t.d2 && q.d2 == t.d1 &&
                is_similar_d(q.c1, (double)q.p1, (double)q.p1 + 1.0, t.c2, (double)t.p2, (double)t.p2 + 1.0, 0.1) && q.c2 == t.c1 && q.d2 == q.d1) {
                const uint64_t key = KEY(slot, 3, (uint64_t)ti * (uint64_t)ntr + (uint64_t)bi);
                if (q.d1 == 0) {
                    const long long sz = (long long)t.p1 - q.p2 + 1;
                    if (MIN <= sz && sz <= MAX) {
                        const long long i_ = (long long)atomicAdd(sig.counter, 1ull);
                        write_sig(sig, i_, key, SVX_DUP_INT, SVX_SRC_SUPPL, 0, q.c2, q.p2, (long long)t.p1 + 1, q.c1, ((long long)q.p1 + 1 + t.p2) / 2, rid, -1, 0, 0);
                    }
                } else {
                    const long long sz = (long long)q.p2 - t.p1;
                    if (MIN <= sz && sz <= MAX) {
                        const long long i_ = (long long)atomicAdd(sig.counter, 1ull);
                        write_sig(sig, i_, key, SVX_DUP_INT, SVX_SRC_SUPPL, 0, q.c2, t.p1, (long long)q.p2 + 1, q.c1, ((long long)q.p1 + t.p2 + 1) / 2, rid, -1, 0, 0);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Ordering + gather
// ------------------------------------------------------------------------------------------------------
__global__ void k_iota_u32(uint32_t* v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

__global__ void k_permute_sigs(SigPtrs in, SigPtrs out, const uint32_t* idx, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { out.qlen[n] = 0; return; }                    // sentinel so that the scan yields the total
    const uint32_t s = idx[i];
    out.type[i] = in.type[s]; out.src[i] = in.src[s]; out.aux[i] = in.aux[s]; out.contig[i] = in.contig[s];
    out.start[i] = in.start[s]; out.end[i] = in.end[s]; out.contig2[i] = in.contig2[s]; out.pos2[i] = in.pos2[s];
    out.read_id[i] = in.read_id[s]; out.rec[i] = in.rec[s]; out.qpos[i] = in.qpos[s]; out.qlen[i] = in.qlen[s];
}

// one wave per signature: unpack qlen 4-bit bases of the owning record into one code per byte
// A batch with sparse SEQ (svx_batch.seq_rng_*: only some ranges of a record's bases are present, see svx_bam_set_seq_filter) is looked up
// through the record's range list; a wanted range that is not there is counted (the caller fails loudly).
__global__ __launch_bounds__(256) void k_gather_seq(SigPtrs s, long long n, const int64_t* seq_off, uint8_t* seq_out,
                                                    const uint64_t* rec_seq_off, const uint8_t* rec_seq, const uint32_t* rng_off, const int32_t* rng_q0,
                                                    const int32_t* rng_len, const uint64_t* rng_byte, unsigned long long* missing) {
    // 16 lanes per signature, four signatures per wave (round 6: a wave per signature - most of them deletions without bases, the insertions ~ 370 bases - spent its
    // time on its chain of dependent loads, signature -> record -> bytes; four chains per wave: 0.33 -> 0.1x ms on configs[1])
    constexpr int GS = 16;
    const int lane = lane_id() & (GS - 1);
    const long long i = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / GS) + (lane_id() / GS);
    if (i >= n) return;
    const int len = s.qlen[i];
    if (len <= 0) return;
    const uint8_t* src;
    uint8_t* dst = seq_out + seq_off[i];
    int q0 = s.qpos[i];
    if (rng_off) {
        const int rec = s.rec[i];
        src = nullptr;
        for (uint32_t r = rng_off[rec]; r < rng_off[rec + 1]; r++) {
            const int a = rng_q0[r];
            if (a <= q0 && q0 + len <= a + rng_len[r]) { src = rec_seq + rng_byte[r]; q0 -= a; break; }      // a is even: nibble parity is kept
        }
        if (!src) { if (lane == 0) atomicAdd(missing, 1ull); return; }
    } else src = rec_seq + rec_seq_off[s.rec[i]];
    // 8 bases per lane and step: one 8-byte load + one 8-byte store while 16 more bases of THIS slice remain (the load then stays inside
    // the record's own bytes), byte accesses for the last chunks
    for (int k0 = lane * 8; k0 < len; k0 += 8 * GS) {
        const int q = q0 + k0;
        if (k0 + 16 <= len) {
            unsigned long long x;
            __builtin_memcpy(&x, src + (q >> 1), 8);
            unsigned long long out = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int nib = (q & 1) + t;                         // nibble index from the first loaded byte, high nibble first
                const unsigned by = (unsigned)(x >> (8 * (nib >> 1))) & 0xffu;
                out |= (unsigned long long)((nib & 1) ? (by & 15u) : (by >> 4)) << (8 * t);
            }
            __builtin_memcpy(dst + k0, &out, 8);
        } else {
            for (int t = 0; t < 8 && k0 + t < len; t++) {
                const uint8_t by = src[(q + t) >> 1];
                dst[k0 + t] = ((q + t) & 1) ? (by & 15) : (by >> 4);
            }
        }
    }
}

static int order_and_store(svx_ctx* c, DevSigs& raw, DevSigs& out, int64_t n) {
    hipStream_t st = c->stream;
    SVXCHK(out.reserve(n + 1));
    out.n = n;
    if (n == 0) return SVX_OK;
    SVXCHK(c->tmp0.reserve((size_t)n * 4));
    SVXCHK(c->tmp1.reserve((size_t)n * 4));
    const int T = 256;
    k_iota_u32<<<(unsigned)((n + T - 1) / T), T, 0, st>>>(c->tmp0.as<uint32_t>(), n);
    SVXCHK(svx_sort_pairs_u64(c, raw.key.as<uint64_t>(), out.key.as<uint64_t>(), c->tmp0.as<uint32_t>(), c->tmp1.as<uint32_t>(), n, 0, 64));
    k_permute_sigs<<<(unsigned)((n + 1 + T - 1) / T), T, 0, st>>>(sig_ptrs(raw), sig_ptrs(out), c->tmp1.as<uint32_t>(), n);
    HIPCHK(hipGetLastError());
    return SVX_OK;
}

int svx_collect_impl(svx_ctx* c, const svx_batch* bd, const svx_params* p) {
    hipStream_t st = c->stream;
    const svx_batch& b = *bd;
    SVXCHK(c->counters.reserve(16 * 8));
    // geometry records of the records and, behind them, of the segment rows (one array: the scan addresses it by item); seg_geom holds the scan's item table
    SVXCHK(c->rec_geom.reserve((size_t)(b.n_rec + b.n_seg + 2) * 5 * 4));
    SVXCHK(c->seg_geom.reserve((size_t)(b.n_rec + b.n_seg + 1) * sizeof(uint4)));
    int* const rec_geom_p = c->rec_geom.as<int>(); int* const seg_geom_p = rec_geom_p + 5 * b.n_rec;
    const size_t ws_n = (size_t)(b.n_seg + b.n_rec + 1);
    SVXCHK(c->seg_ws.reserve(ws_n * (sizeof(ASeg) + sizeof(TDup) + sizeof(Trn))));
    ASeg* ws_al = c->seg_ws.as<ASeg>();
    TDup* ws_td = reinterpret_cast<TDup*>(ws_al + ws_n);
    Trn* ws_tr = reinterpret_cast<Trn*>(ws_td + ws_n);
    // total op counts (needed to bound the 16-byte loads at the very end of the arrays)
    uint64_t tot_ops = 0, tot_seg_ops = 0;
    if (b.n_seg > 0) SVXCHK(svx_mail_read2(c, st, b.cigar_off + b.n_rec, 1, &tot_ops, b.seg_cigar_off + b.n_seg, 1, &tot_seg_ops));
    else SVXCHK(svx_mail_read(c, st, b.cigar_off + b.n_rec, 1, &tot_ops));
    int64_t cap_sig = c->raw_sig.cap > 0 ? c->raw_sig.cap : 0, cap_bnd = c->raw_bnd.cap > 0 ? c->raw_bnd.cap : 0;
    int64_t want_sig = (int64_t)(tot_ops / 256) + 4 * b.n_seg + 16 * (int64_t)RAW_SHARDS;      // >= 16 raw slots per wave-private region
    if (cap_sig < want_sig) cap_sig = want_sig;
    if (cap_bnd < 1024) cap_bnd = 1024;
    if (p->all_bnds && cap_bnd < want_sig) cap_bnd = want_sig;
    unsigned long long h_cnt[16];
    for (int attempt = 0; attempt < 3; attempt++) {
        SVXCHK(c->raw_sig.reserve(cap_sig));
        SVXCHK(c->raw_bnd.reserve(cap_bnd));
        HIPCHK(hipMemsetAsync(c->counters.p, 0, 16 * 8, st));
        EmitTarget ts{sig_ptrs(c->raw_sig), c->raw_sig.cap, c->counters.as<unsigned long long>() + CNT_SIG};
        EmitTarget tb{sig_ptrs(c->raw_bnd), c->raw_bnd.cap, c->counters.as<unsigned long long>() + CNT_BND};
        HIPCHK(hipEventRecord(c->ev[0], st));
        const long long items = b.n_rec + b.n_seg;
        if (items > 0) {
            long long blocks = (items + 3) / 4;
            if (items >= (1ll << 31)) return svx_fail(SVX_E_ARG, "more than 2^31 records + segment rows in one batch", __FILE__, __LINE__, hipSuccess);
            static int per_cu = 0;                                         // resident 256-thread blocks per CU for this kernel
            { const char* f = getenv("SVX_SCAN_BLOCKS"); if (f) per_cu = atoi(f); }
            if (!per_cu) {
                int occ = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_cigar_scan, 256, 0) != hipSuccess || occ < 1) occ = 4;
                per_cu = occ > 8 ? 8 : occ;                                // 8 blocks = 8 waves per SIMD: the kernel is built for <= 64 VGPRs / <= 80 SGPRs (Makefile check)
            }
            long long max_blocks = (long long)c->n_cu * per_cu;
            if (max_blocks > RAW_SHARDS / 4) max_blocks = RAW_SHARDS / 4;
            if (blocks > max_blocks) blocks = max_blocks;
            const long long shard_cap = (c->raw_sig.cap + RAW_SHARDS - 1) / RAW_SHARDS;
            SVXCHK(c->raw_indel.reserve((size_t)shard_cap * RAW_SHARDS * sizeof(RawIndel)));
            SVXCHK(c->shard_cnt.reserve((size_t)RAW_SHARDS * 8 + (size_t)(RAW_SHARDS + 1) * 8));
            unsigned long long* shard_counter = c->shard_cnt.as<unsigned long long>();
            long long* shard_prefix = reinterpret_cast<long long*>(shard_counter + RAW_SHARDS);
            const bool zero_in_prepare = items >= RAW_SHARDS;                 // (a grid of at least RAW_SHARDS threads)
            if (!zero_in_prepare) HIPCHK(hipMemsetAsync(shard_counter, 0, RAW_SHARDS * 8, st));
            RawTarget rt{c->raw_indel.as<RawIndel>(), shard_cap, shard_counter};
            // the scan's 16-byte loads need an array of at least four operations (load_chunk): a smaller one is copied in front of no-op padding
            const uint32_t* cigar_p = b.cigar; uint64_t tot_scan = tot_ops;
            if (tot_ops < 4 && b.n_rec > 0) {
                SVXCHK(c->tmp5.reserve(64));
                HIPCHK(hipMemsetAsync(c->tmp5.p, 0x0f, 16, st));                      // op 15: advances nothing, reports nothing
                if (tot_ops) HIPCHK(hipMemcpyAsync(c->tmp5.p, b.cigar, (size_t)tot_ops * 4, hipMemcpyDeviceToDevice, st));
                cigar_p = c->tmp5.as<uint32_t>(); tot_scan = 4;
            }
            ScanArgs sa{b.n_rec, b.n_seg, b.flag, b.mapq, b.lseq, b.seg_off, b.cigar_off, cigar_p, b.seg_lseq, b.seg_cigar_off, b.seg_cigar, p->min_mapq, p->min_sv_size};
            k_scan_prepare<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(sa, tot_scan, tot_seg_ops, c->seg_geom.as<uint4>(), rec_geom_p, zero_in_prepare ? shard_counter : nullptr);
            k_cigar_scan<<<(unsigned)blocks, 256, 0, st>>>(c->seg_geom.as<uint4>(), (uint32_t)b.n_rec, (uint32_t)items, cigar_p, b.seg_cigar, p->min_sv_size, rt, rec_geom_p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev[5], st));
            // dense slots for the raw records; the total becomes the start of the segment kernel's allocations
            k_shard_prefix<<<1, 1024, 0, st>>>(shard_counter, shard_cap, shard_prefix, c->counters.as<unsigned long long>());
            k_emit_indels<<<dim3((unsigned)((shard_cap + 255) / 256), RAW_SHARDS), 256, 0, st>>>(b, *p, c->raw_indel.as<RawIndel>(), shard_counter, shard_cap,
                                                                                              shard_prefix, ts, tb);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(c->ev[1], st));
        if (b.n_rec > 0 && b.n_seg > 0) {
            k_segments<<<(unsigned)((b.n_rec + 255) / 256), 256, 0, st>>>(b, *p, ts, tb, rec_geom_p, seg_geom_p, ws_al, ws_td, ws_tr);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(c->ev[2], st));
        if (b.n_rec > 0) k_count_used<<<(unsigned)std::min<long long>((b.n_rec + 255) / 256, 1024), 256, 0, st>>>(b, *p, c->counters.as<unsigned long long>());
        SVXCHK(svx_mail_read(c, st, c->counters.p, 16, h_cnt));
        if ((int64_t)h_cnt[CNT_SIG] <= c->raw_sig.cap && (int64_t)h_cnt[CNT_BND] <= c->raw_bnd.cap && h_cnt[CNT_OVERFLOW] == 0) break;
        if (attempt == 2) return svx_fail(SVX_E_CAPACITY, "signature buffers", __FILE__, __LINE__, hipSuccess);
        cap_sig = 2 * ((int64_t)h_cnt[CNT_SIG] + (int64_t)h_cnt[CNT_OVERFLOW]) + 64 * (int64_t)RAW_SHARDS; cap_bnd = (int64_t)h_cnt[CNT_BND] + 1024;
        if (cap_sig < 4 * c->raw_sig.cap && h_cnt[CNT_OVERFLOW]) cap_sig = 4 * c->raw_sig.cap;     // shard imbalance: grow generously
    }
    const int64_t n_sig = (int64_t)h_cnt[CNT_SIG], n_bnd = (int64_t)h_cnt[CNT_BND];
    SVXCHK(order_and_store(c, c->raw_sig, c->sig, n_sig));
    SVXCHK(order_and_store(c, c->raw_bnd, c->bnd, n_bnd));
    HIPCHK(hipEventRecord(c->ev[3], st));
    // inserted sequences of the main list
    SVXCHK(c->sig.seq_off.reserve((size_t)(n_sig + 2) * 8));
    int64_t n_seq = 0;
    if (n_sig > 0 && !c->no_seq_gather) {
        SVXCHK(svx_exclusive_scan_i32_to_i64(c, c->sig.qlen.as<int32_t>(), c->sig.seq_off.as<int64_t>(), n_sig + 1));
        SVXCHK(svx_mail_read(c, st, c->sig.seq_off.as<int64_t>() + n_sig, 1, &n_seq));
        SVXCHK(c->sig.seq.reserve((size_t)n_seq + 16));
        k_gather_seq<<<(unsigned)((n_sig + 15) / 16), 256, 0, st>>>(sig_ptrs(c->sig), n_sig, c->sig.seq_off.as<int64_t>(), c->sig.seq.as<uint8_t>(),
                                                                 b.seq_off, b.seq, b.seq_rng_off, b.seq_rng_q0, b.seq_rng_len, b.seq_rng_byte,
                                                                 c->counters.as<unsigned long long>() + CNT_SEQ_MISSING);
        HIPCHK(hipGetLastError());
        if (b.seq_rng_off) {
            unsigned long long miss = 0;
            SVXCHK(svx_mail_read(c, st, c->counters.as<unsigned long long>() + CNT_SEQ_MISSING, 1, &miss));
            if (miss) return svx_fail(SVX_E_ARG, "sparse SEQ: the bases of a reported insertion are not in the batch (svx_bam_set_seq_filter larger than params.min_sv_size?)",
                                      __FILE__, __LINE__, hipSuccess);
        }
    } else {
        HIPCHK(hipMemsetAsync(c->sig.seq_off.p, 0, (size_t)(n_sig + 2) * 8, st));
        SVXCHK(c->sig.seq.reserve(16));
    }
    c->sig.n_seq = n_seq;
    // the side list carries no sequences
    SVXCHK(c->bnd.seq_off.reserve((size_t)(n_bnd + 2) * 8));
    HIPCHK(hipMemsetAsync(c->bnd.seq_off.p, 0, (size_t)(n_bnd + 2) * 8, st));
    SVXCHK(c->bnd.seq.reserve(16));
    c->bnd.n_seq = 0;
    HIPCHK(hipEventRecord(c->ev[4], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms;
    svx_stats& s = c->stats;
    if (b.n_rec + b.n_seg > 0) { HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[5])); s.t_cigar_scan_ms = ms; } else s.t_cigar_scan_ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); s.t_segments_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3])); s.t_sort_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[3], c->ev[4])); s.t_gather_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[4])); s.t_collect_ms = ms;
    s.n_rec_used = (int64_t)h_cnt[CNT_USED]; s.n_ops = (int64_t)h_cnt[CNT_OPS]; s.n_seg = b.n_seg; s.n_seg_ops = (int64_t)tot_seg_ops;
    s.n_sig = n_sig; s.n_bnd_side = n_bnd; s.n_ins_bases = n_seq;
    return SVX_OK;
}

// loads this translation unit's code object (HIP does it lazily, at the first launch): called by svx_ctx_create so that the first COLLECT / CLUSTER call
// of a context does not pay for it
void svx_preload_collect() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_shard_prefix)); (void)hipGetLastError(); }
