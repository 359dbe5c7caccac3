// cluster.hip - CLUSTER on the GPU: partition -> sample -> pairwise distance -> average linkage -> flat cut ->
// consolidate.
//
// Reference being replaced (eldariont/svim v2.0.0):
//   cluster_sv_signatures                      src/svim/SVIM_CLUSTER.py:7-26
//   partition_and_cluster / form_partitions    src/svim/SVIM_clustering.py:375-385 / :17-29
//   clusters_from_partitions                   :122-180  (seed(1524), sample(...,100), same-read dedupe, 99999 rule)
//   span_position_distance                     :47-96
//   scipy linkage('average') + fcluster('distance')   call sites :170-171 (nn_chain / label / cluster_dist)
//   consolidate_clusters_unilocal/_bilocal, calculate_score   :183-303
//   random.sample / MT19937                    CPython Lib/random.py, Modules/_randommodule.c
//
// Layout: signatures stay in their SoA table; one 64-bit + one 32-bit radix key order them exactly like
// sorted(key=get_key) (stable, contig names compared through their precomputed string rank).  One wavefront
// owns one partition (<= 100 sampled signatures): member records, the condensed FP64 distance matrix
// (<= 4950 doubles = 39.6 KB) and the whole dendrogram live in LDS; nothing is re-read from HBM.
// FP64 throughout with the reference's operation order (-ffp-contract=off), so labels are bit-exact.
#include "common.hpp"
#include "hostcopy.hpp"
#include <cstdlib>

int svx_launch_edit_pairs(svx_ctx* c, int64_t n_work, const void* work_dev, const ClusterIn& in, int32_t* ed_dev,
                          unsigned long long* cells_dev);

struct EditWork { uint32_t a, b; long long slot; };

#define MAXN 100

// ---------------------------------------------------------------------------------------------------------
// sort keys (get_key): hi = type | rank1 | rank2, lo = biased coordinate
// ---------------------------------------------------------------------------------------------------------
__global__ void k_make_keys(ClusterIn in, const int32_t* rank, uint64_t* hi, uint64_t* lo, uint32_t* idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= in.n) return;
    const int t = in.type[i];
    uint64_t r1, r2 = 0; int32_t coord;
    if (t == SVX_INS) { r1 = (uint64_t)rank[in.contig[i]]; coord = in.start[i]; }
    else if (t == SVX_DUP_INT) { r1 = (uint64_t)rank[in.contig2[i]]; r2 = (uint64_t)rank[in.contig[i]]; coord = in.pos2[i]; }
    else if (t == SVX_BND) { r1 = (uint64_t)rank[in.contig[i]]; coord = in.start[i]; }
    else { r1 = (uint64_t)rank[in.contig[i]]; coord = in.end[i]; }
    hi[i] = ((uint64_t)t << 56) | (r1 << 28) | r2;
    lo[i] = (uint64_t)((uint32_t)coord ^ 0x80000000u);
    idx[i] = (uint32_t)i;
}

__global__ void k_iota_u32c(uint32_t* v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

__global__ void k_gather_u64(const uint64_t* src, const uint32_t* idx, uint64_t* dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// new-partition flags in sorted order (form_partitions: gap to the PREVIOUS element only)
__global__ void k_part_flags(ClusterIn in, const uint64_t* hi_sorted, const uint32_t* sidx, long long max_distance, int64_t* flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > in.n) return;
    if (i == in.n) { flag[i] = 0; return; }
    int f = 1;
    if (i > 0 && hi_sorted[i] == hi_sorted[i - 1]) {          // same type and same contig(s)
        const uint32_t a = sidx[i - 1], b = sidx[i];
        const int t = in.type[b];
        long long d;
        if (t == SVX_INS) d = (long long)in.start[b] - in.start[a];
        else if (t == SVX_DUP_INT) d = (long long)in.pos2[b] - in.pos2[a];
        else d = (long long)in.start[b] - in.end[a];
        if (d < 0) d = 0;
        f = d > max_distance;
    }
    flag[i] = f;
}

// part_start[pid] = first sorted position of partition pid
__global__ void k_part_starts(const int64_t* flag, const int64_t* pid_excl, long long n, int64_t* part_start, long long n_part) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { part_start[n_part] = n; return; }
    if (flag[i]) part_start[pid_excl[i]] = i;
}

// per-partition derived sizes: ns = min(size, 100); large flag; INS pair slots
__global__ void k_part_sizes(const int64_t* part_start, long long n_part, const uint8_t* type, const uint32_t* sidx,
                             int64_t* ns, int64_t* large, int64_t* pairs) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n_part) return;
    if (p == n_part) { ns[p] = 0; large[p] = 0; pairs[p] = 0; return; }
    const long long size = part_start[p + 1] - part_start[p];
    const long long m = size > MAXN ? MAXN : size;
    ns[p] = m;
    large[p] = size > MAXN;
    pairs[p] = type[sidx[part_start[p]]] == SVX_INS ? m * (m - 1) / 2 : 0;
}

__global__ void k_large_list(const int64_t* large, const int64_t* large_excl, long long n_part, int32_t* list) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_part && large[p]) list[large_excl[p]] = (int32_t)p;
}

// ---------------------------------------------------------------------------------------------------------
// random.sample(partition, 100) for every partition > 100, one wave per signature type (the RNG is re-seeded per
// type and carried across that type's partitions: SVIM_clustering.py:129-134)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

// The tempered MT19937 words after random.seed(1524) are the same for every type and every call: they are generated ONCE per
// context (k_mt_generate, regenerated only when a call needs a longer prefix) and every consumer just reads them.
__global__ __launch_bounds__(64) void k_mt_generate(const uint32_t* mt_init, uint32_t* words, long long n_blocks) {
    __shared__ uint32_t s[624];
    const int lane = lane_id();
    for (int i = lane; i < 624; i += 64) s[i] = mt_init[i];
    __syncthreads();
    for (long long blk = 0; blk < n_blocks; blk++) {
        // three dependency phases; inside a phase every lane reads before any lane of the same instruction writes
        for (int k0 = 0; k0 < 227; k0 += 64) { const int k = k0 + lane; uint32_t v = 0; const bool ok = k < 227;
            if (ok) { const uint32_t y = (s[k] & 0x80000000u) | (s[k + 1] & 0x7fffffffu); v = s[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            __syncthreads(); if (ok) s[k] = v; __syncthreads(); }
        for (int k0 = 227; k0 < 454; k0 += 64) { const int k = k0 + lane; uint32_t v = 0; const bool ok = k < 454;
            if (ok) { const uint32_t y = (s[k] & 0x80000000u) | (s[k + 1] & 0x7fffffffu); v = s[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            __syncthreads(); if (ok) s[k] = v; __syncthreads(); }
        for (int k0 = 454; k0 < 623; k0 += 64) { const int k = k0 + lane; uint32_t v = 0; const bool ok = k < 623;
            if (ok) { const uint32_t y = (s[k] & 0x80000000u) | (s[k + 1] & 0x7fffffffu); v = s[k - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            __syncthreads(); if (ok) s[k] = v; __syncthreads(); }
        if (lane == 0) { const uint32_t y = (s[623] & 0x80000000u) | (s[0] & 0x7fffffffu); s[623] = s[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
        __syncthreads();
        for (int k = lane; k < 624; k += 64) words[blk * 624 + k] = mt_temper(s[k]);
    }
}

// cursor of one signature type over the shared word stream: lane i of `buf` holds word cur + i.  The words are staged through LDS
// 4096 at a time: the walk is one lone wave, which cannot hide a dependent HBM/L2 round trip (~2 us) per 64 words.
#define MT_STAGE 4096
struct MtStream {
    const uint32_t* words;  // tempered words, generation order
    uint32_t* stage;        // LDS, MT_STAGE words
    long long cap;          // words available
    long long cur, sbase;   // global index of buf's lane 0 / of stage[0]
    uint32_t buf;
    int pos, lim;           // uniform cursor inside buf
    int overflow;
    __device__ void start() { cur = -64; sbase = -(long long)MT_STAGE; pos = 0; lim = 0; overflow = 0; buf = 0; }
    __device__ long long position() const { return cur + pos; }      // global index of the next word
    __device__ void refill() {
        cur += 64;
        if (cur + 64 > cap) overflow = 1;
        if (cur + 64 > sbase + MT_STAGE) {
            sbase = cur;
            uint32_t v[MT_STAGE / 64];
#pragma unroll
            for (int i = 0; i < MT_STAGE / 64; i++) { const long long k = sbase + i * 64 + lane_id(); v[i] = (k < cap) ? words[k] : 0u; }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT_STAGE / 64; i++) stage[i * 64 + lane_id()] = v[i];
            __syncthreads();
        }
        buf = stage[(int)(cur - sbase) + lane_id()];
        lim = 64; pos = 0;
    }
    __device__ uint32_t next() {
        if (pos == lim) refill();
        const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)buf, pos);
        pos++;
        return w;
    }
    // Consume exactly the words random.sample(range(n), 100) consumes with the pool method (n <= 1045) WITHOUT producing
    // the sample: draw i accepts a word iff word >> (32-k) < n-i.  64 words are classified at once; the only coupling
    // between lanes is the number of earlier accepts, resolved by a short monotone fix-point on ballots.
    __device__ void skip_pool_sample(uint32_t n) {
        const int lane = lane_id();
        int i = 0;
        while (i < 100) {
            if (pos == lim) refill();
            const uint32_t bound = n - (uint32_t)i;
            const int k = 32 - __clz((int)bound);
            // k stays valid while the bound has the same bit length: at most `cap` accepts in this round
            const int cap = (int)(bound - (1u << (k - 1))) + 1;
            int need = 100 - i;
            if (cap < need) need = cap;
            const bool valid = lane >= pos && lane < lim;
            const unsigned long long V = __ballot(valid);
            const uint32_t r = buf >> (32 - k);
            unsigned long long A = __ballot(valid && r + (uint32_t)__popcll(V & lanemask_lt()) < bound);    // accepted whatever happened before
            unsigned long long R = __ballot(valid && r >= bound);                                          // rejected whatever happened before
            unsigned long long U = V & ~A & ~R;
            while (U) {
                const uint32_t a_lo = (uint32_t)__popcll(A & lanemask_lt()), a_hi = (uint32_t)__popcll((A | U) & lanemask_lt());
                const bool inU = (U >> lane) & 1ull;
                // signed: the upper estimate a_hi may exceed a small bound (then the word cannot be accepted)
                const unsigned long long nA = __ballot(inU && (int)r < (int)bound - (int)a_hi), nR = __ballot(inU && (int)r >= (int)bound - (int)a_lo);
                A |= nA; R |= nR; U &= ~(nA | nR);
            }
            const int c = __popcll(A);
            if (c < need) { i += c; pos = lim; }
            else {
                const bool inA = (A >> lane) & 1ull;
                const unsigned long long hit = __ballot(inA && (int)__popcll(A & lanemask_lt()) + 1 == need);
                pos = __ffsll((long long)hit);           // one past the lane holding the need-th accepted word
                i += need;
            }
        }
    }
    __device__ uint32_t randbelow(uint32_t n) {      // Random._randbelow_with_getrandbits
        const int k = 32 - __clz((int)n);
        uint32_t r = next() >> (32 - k);
        while (r >= n) r = next() >> (32 - k);
        return r;
    }
};

// Phase A: one wave per type walks that type's large partitions IN ORDER and only decides how many words each one
// consumes (pool method, n <= 1045: acceptance does not depend on the sampled values).  Partitions above 1045 use the
// set method, whose rejections depend on the values drawn: they are sampled right here.
__global__ __launch_bounds__(64) void k_sample_scan(const int32_t* large_list, long long n_large, const int64_t* part_start, const uint32_t* sidx,
                                                    const uint8_t* type, const int64_t* large_excl, const uint32_t* stream,
                                                    long long cap, long long* samp_start, int32_t* sample_idx, int* err,
                                                    long long* chain /* [2 NTYPES]: stream position where each type starts / (out) stops */) {
    const int t = blockIdx.x, lane = lane_id();
    if (lane == 0) chain[SVX_NTYPES + t] = chain[t];
    // range of large partitions whose type is t (types are non-decreasing along the sorted order)
    long long lo = 0, hi = n_large;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((int)type[sidx[part_start[large_list[mid]]]] < t) lo = mid + 1; else hi = mid; }
    const long long begin = lo;
    hi = n_large;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((int)type[sidx[part_start[large_list[mid]]]] <= t) lo = mid + 1; else hi = mid; }
    const long long end = lo;
    if (begin == end) return;
    __shared__ uint32_t stage[MT_STAGE];
    MtStream mt; mt.words = stream; mt.stage = stage; mt.cap = cap; mt.start();
    mt.cur += chain[t];                                  // this rank continues the type's stream where the previous one stopped
    mt.refill();
    int p_lane = 0; uint32_t n_lane = 0;                 // partition ids / sizes of 64 list entries at a time: no dependent loads in the walk
    for (long long q = begin; q < end; q++) {
        const int qi = (int)((q - begin) & 63);
        if (qi == 0) {
            const long long ql = q + lane;
            p_lane = ql < end ? large_list[ql] : 0;
            n_lane = ql < end ? (uint32_t)(part_start[p_lane + 1] - part_start[p_lane]) : 0u;
        }
        const int p = __builtin_amdgcn_readlane(p_lane, qi);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)n_lane, qi);
        if (n <= 1045) {
            if (lane == 0) samp_start[q] = mt.position();
            mt.skip_pool_sample(n);
        } else {
            if (lane == 0) samp_start[q] = -1;
            uint32_t res_a = 0, res_b = 0;
            for (int i = 0; i < 100; i++) {
                uint32_t j;
                for (;;) {
                    j = mt.randbelow(n);
                    const unsigned long long seen = __ballot((lane < i && res_a == j) || (lane < i - 64 && res_b == j));
                    if (!seen) break;
                }
                if (i < 64) { if (lane == i) res_a = j; }
                else if (lane == i - 64) res_b = j;
            }
            int32_t* out = sample_idx + large_excl[p] * 100;
            out[lane] = (int32_t)res_a;
            if (lane < 36) out[64 + lane] = (int32_t)res_b;
        }
    }
    if (lane == 0) chain[SVX_NTYPES + t] = mt.position();
    if (mt.overflow && lane == 0) *err = 1;
}

// ---- Phase A without the serial walk -----------------------------------------------------------------------------------
// How many words a pool-method partition consumes depends only on its size and on where in the stream it starts.  The host
// brackets every partition's start (window = expected start -+ 6 sigma, from the exact mean / variance of the rejection
// sampling), k_sample_tables simulates EVERY (partition, candidate start) with one lane each - ~10^6 independent ~150-word
// walks, ideal GPU work - and following the chain is one table lookup per partition (k_chase_*).  Any surprise (a start outside its
// window, a set-method partition, the stream running out) raises `err` and the caller falls back to k_sample_scan.
struct SampleMeta { long long lo; int width; int n; long long off; };      // window [lo, lo + width), partition size, first table slot

__global__ void k_large_info(const int32_t* large_list, long long n_large, const int64_t* part_start, const uint32_t* sidx, const uint8_t* type,
                             int32_t* info /* [2 n_large]: type, size */) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_large) return;
    const int p = large_list[q];
    info[2 * q] = type[sidx[part_start[p]]];
    info[2 * q + 1] = (int32_t)(part_start[p + 1] - part_start[p]);
}

// one thread per candidate start of a partition: how many stream words do its 100 draws consume from there?  The 256 starts of a block are
// consecutive, so the words they will look at (a window of 256 + the ~130-250 a walk consumes) are staged in LDS once: the walk is a chain of
// dependent reads, ~10x shorter from LDS than from L2; the rare walk that leaves the window continues in global memory.
#define SAMPLE_WIN 1024
__global__ __launch_bounds__(256) void k_sample_tables(const SampleMeta* meta, const uint32_t* stream, long long cap, uint16_t* table) {
    __shared__ uint32_t win[SAMPLE_WIN];
    const SampleMeta m = meta[blockIdx.y];
    if ((int)blockIdx.x * 256 >= m.width) return;
    const long long base = m.lo + (long long)blockIdx.x * 256;
    for (int i = threadIdx.x; i < SAMPLE_WIN; i += 256) { const long long q = base + i; win[i] = q < cap ? stream[q] : 0u; }
    __syncthreads();
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= m.width) return;
    long long pos = m.lo + s;
    const long long start = pos;
    uint32_t bound = (uint32_t)m.n;
    bool ok = true;
    for (int i = 0; i < 100; i++, bound--) {
        const int k = 32 - __clz((int)bound);           // Random._randbelow_with_getrandbits: k = bound.bit_length()
        for (;;) {
            if (pos >= cap) { ok = false; break; }
            const long long o = pos - base;
            const uint32_t word = o < SAMPLE_WIN ? win[o] : stream[pos];
            pos++;
            if ((word >> (32 - k)) < bound) break;
        }
        if (!ok) break;
    }
    const long long used = pos - start;
    table[m.off + s] = (ok && used < 0xffff) ? (uint16_t)used : (uint16_t)0xffff;
}

// Following the tables is one dependent lookup per partition (0.55 us each for a lone lane).  The chain is therefore cut into runs of
// CHASE_RUN partitions: (1) every candidate start of a run's first partition is followed to the end of the run, all in parallel;
// (2) one lane per type hops from run to run; (3) one lane per run fills in the starts inside it.  Serial depth: 2 CHASE_RUN + runs.
#define CHASE_RUN 64
struct ChaseRun { long long first, last; long long eoff; int type; int pad; };      // partitions [first, last), slot of its first candidate in `ends`

__global__ __launch_bounds__(256) void k_chase_runs(const SampleMeta* meta, const ChaseRun* runs, const uint16_t* table, long long* ends) {
    const ChaseRun r = runs[blockIdx.y];
    const SampleMeta m0 = meta[r.first];
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= m0.width) return;
    long long pos = m0.lo + s;
    for (long long q = r.first; q < r.last; q++) {
        const SampleMeta m = meta[q];
        const long long idx = pos - m.lo;
        if (idx < 0 || idx >= m.width) { pos = -1; break; }
        const uint16_t used = table[m.off + idx];
        if (used == 0xffff) { pos = -1; break; }
        pos += used;
    }
    ends[r.eoff + s] = pos;                                   // stream position after the run, -1: left the windows
}

__global__ void k_chase_top(const SampleMeta* meta, const ChaseRun* runs, const long long* type_run_begin /* [NTYPES + 1] */, const long long* ends,
                            long long* run_start, int* err, long long* chain /* [2 NTYPES]: start / (out) end position per type */) {
    const int t = blockIdx.x;
    if (threadIdx.x != 0) return;
    long long pos = chain[t];
    chain[SVX_NTYPES + t] = pos;
    for (long long b = type_run_begin[t]; b < type_run_begin[t + 1]; b++) {
        const ChaseRun r = runs[b];
        const SampleMeta m0 = meta[r.first];
        const long long idx = pos - m0.lo;
        run_start[b] = pos;
        if (idx < 0 || idx >= m0.width) { *err = 1; return; }
        pos = ends[r.eoff + idx];
        if (pos < 0) { *err = 1; return; }
    }
    chain[SVX_NTYPES + t] = pos;
}

__global__ void k_chase_fill(const SampleMeta* meta, const ChaseRun* runs, long long n_runs, const uint16_t* table, const long long* run_start,
                             long long* samp_start) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_runs) return;
    const ChaseRun r = runs[b];
    long long pos = run_start[b];
    for (long long q = r.first; q < r.last; q++) {
        const SampleMeta m = meta[q];
        const long long idx = pos - m.lo;
        if (idx < 0 || idx >= m.width) return;                // only after k_chase_top raised err
        samp_start[q] = pos;
        pos += table[m.off + idx];
    }
}

// Phase B: one wave per pool-method partition replays random.sample from its slice of the word stream.
__global__ __launch_bounds__(64) void k_sample_apply(const int32_t* large_list, long long n_large, const int64_t* part_start, const uint32_t* sidx,
                                                     const uint8_t* type, const int64_t* large_excl, const uint32_t* stream, long long cap_per_type,
                                                     const long long* samp_start, int32_t* sample_idx) {
    const long long q = blockIdx.x;
    if (q >= n_large) return;
    const long long st = samp_start[q];
    if (st < 0) return;
    const int lane = lane_id();
    const int p = large_list[q];
    const uint32_t* strm = stream;              // one stream: every type restarts from seed(1524)
    const uint32_t n = (uint32_t)(part_start[p + 1] - part_start[p]);
    long long chunk = st;                   // words [chunk, chunk+64) sit in `buf`
    uint32_t buf = (chunk + lane < cap_per_type) ? strm[chunk + lane] : 0u;
    int pos = 0;
    // per-step records kept in registers: lane l holds steps l (a) and 64+l (b)
    uint32_t pos_a = 0, val_a = 0, res_a = 0, pos_b = 0, val_b = 0, res_b = 0;
    // pool method: result[i] = pool[j]; pool[j] = pool[n-i-1]; the pool is virtual (identity + <= 100 overrides)
    for (int i = 0; i < 100; i++) {
        const uint32_t bound = n - (uint32_t)i;
        const int k = 32 - __clz((int)bound);
        uint32_t j;
        for (;;) {
            if (pos == 64) { chunk += 64; buf = (chunk + lane < cap_per_type) ? strm[chunk + lane] : 0u; pos = 0; }
            j = (uint32_t)__builtin_amdgcn_readlane((int)buf, pos) >> (32 - k);
            pos++;
            if (j < bound) break;
        }
        const uint32_t tail = bound - 1u;
        uint32_t vj = j, vt = tail;
        // latest override wins: steps 64.. first, then 0..63
        unsigned long long mb = __ballot(lane < i - 64 && pos_b == j), ma = __ballot(lane < i && pos_a == j);
        if (mb) vj = (uint32_t)__shfl((int)val_b, 63 - __clzll((long long)mb), 64);
        else if (ma) vj = (uint32_t)__shfl((int)val_a, 63 - __clzll((long long)ma), 64);
        mb = __ballot(lane < i - 64 && pos_b == tail); ma = __ballot(lane < i && pos_a == tail);
        if (mb) vt = (uint32_t)__shfl((int)val_b, 63 - __clzll((long long)mb), 64);
        else if (ma) vt = (uint32_t)__shfl((int)val_a, 63 - __clzll((long long)ma), 64);
        if (i < 64) { if (lane == i) { pos_a = j; val_a = vt; res_a = vj; } }
        else if (lane == i - 64) { pos_b = j; val_b = vt; res_b = vj; }
    }
    int32_t* out = sample_idx + large_excl[p] * 100;
    out[lane] = (int32_t)res_a;
    if (lane < 36) out[64 + lane] = (int32_t)res_b;
}

// ---------------------------------------------------------------------------------------------------------
// distances
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long floordiv2(long long x) { return (x >= 0) ? x / 2 : -((-x + 1) / 2); }
__device__ __forceinline__ long long labs64(long long x) { return x < 0 ? -x : x; }

struct Member { int start, end, pos2, read, aux, gidx, c2, pad; };

// span_position_distance (SVIM_clustering.py:47-96); ed = haplotype edit distance for INS pairs that need it
__device__ __forceinline__ double span_position_distance(int t, const Member& a, const Member& b, const svx_params& p, int ed) {
    if (t == SVX_BND) {
        const long long d1 = labs64((long long)a.start - b.start), d2 = labs64((long long)a.pos2 - b.pos2);
        return (a.aux == b.aux) ? (double)(d1 + d2) / 3000.0 : 99999.0;
    }
    const long long span1 = (long long)a.end - a.start, span2 = (long long)b.end - b.start;
    const long long mx = span1 > span2 ? span1 : span2;
    if (t == SVX_INS) {
        const double pd = (double)labs64((long long)a.start - b.start) / p.position_distance_normalizer;
        if (pd > 2 * p.cluster_max_distance) return pd + (double)labs64(span1 - span2) / (double)mx;
        return pd + (double)ed / (double)mx / p.edit_distance_normalizer;
    }
    const long long c1 = floordiv2((long long)a.start + a.end), c2 = floordiv2((long long)b.start + b.end);
    const double pd = (double)labs64(c1 - c2) / p.position_distance_normalizer;
    const double sd = (double)labs64(span1 - span2) / (double)mx;
    if (t == SVX_DUP_INT) {
        const double pdd = (double)labs64((long long)a.pos2 - b.pos2) / p.position_distance_normalizer;
        return pd + pdd + sd;
    }
    return pd + sd;
}

__device__ __forceinline__ bool ins_needs_edit(const Member& a, const Member& b, const svx_params& p) {
    const double pd = (double)labs64((long long)a.start - b.start) / p.position_distance_normalizer;
    return !(pd > 2 * p.cluster_max_distance);
}

__device__ __forceinline__ int cidx(int n, int i, int j) { if (i > j) { const int t = i; i = j; j = t; } return n * i - i * (i + 1) / 2 + (j - i - 1); }

// sampled member q of partition p -> global signature index
__device__ __forceinline__ uint32_t member_gidx(long long pstart, long long size, int q, const uint32_t* sidx, const int32_t* sample_idx,
                                                const int64_t* large_excl, long long p) {
    if (size > MAXN) return sidx[pstart + sample_idx[large_excl[p] * 100 + q]];
    return sidx[pstart + q];
}

// enumerate the INS pairs that need an edit distance (one wave per INS partition)
__global__ __launch_bounds__(64) void k_ins_pairs(long long n_part, const int64_t* part_start, const uint32_t* sidx, const int32_t* sample_idx,
                                                  const int64_t* large_excl, const int64_t* pair_cnt, const int64_t* pair_off, ClusterIn in,
                                                  svx_params p, EditWork* work, unsigned long long* n_work, long long work_cap) {
    const long long pt = blockIdx.x;
    if (pt >= n_part || pair_cnt[pt] == 0) return;
    __shared__ int m_start[MAXN]; __shared__ uint32_t m_g[MAXN];
    const long long ps = part_start[pt], size = part_start[pt + 1] - ps;
    const int ns = size > MAXN ? MAXN : (int)size;
    for (int q = lane_id(); q < ns; q += 64) {
        const uint32_t g = member_gidx(ps, size, q, sidx, sample_idx, large_excl, pt); m_g[q] = g; m_start[q] = in.start[g];
    }
    __syncthreads();
    const int npairs = ns * (ns - 1) / 2;
    const long long base = pair_off[pt];
    const double two_n1 = 2.0 * ns - 1.0;
    // pass 1: which pairs need an edit distance (one ballot mask per 64 condensed indices, kept in LDS); pass 2: write them after ONE
    // atomic for the whole partition - a returning atomic per step costs a lone wave ~2 us each
    __shared__ unsigned long long want_mask[(MAXN * (MAXN - 1) / 2 + 63) / 64];
    int total = 0;
    for (int k0 = 0; k0 < npairs; k0 += 64) {
        const int k = k0 + lane_id();
        bool want = false;
        if (k < npairs) {
            // invert the condensed index k = i*ns - i*(i+1)/2 + (j-i-1): closed form, then one step of correction either way
            int i = (int)((two_n1 - sqrt(two_n1 * two_n1 - 8.0 * k)) * 0.5);
            if (i < 0) i = 0;
            if (i > ns - 2) i = ns - 2;
            while (i > 0 && i * ns - i * (i + 1) / 2 > k) i--;
            while (i < ns - 2 && (i + 1) * ns - (i + 1) * (i + 2) / 2 <= k) i++;
            const int j = i + 1 + (k - (i * ns - i * (i + 1) / 2));
            const double pd = (double)labs64((long long)m_start[i] - m_start[j]) / p.position_distance_normalizer;
            want = !(pd > 2 * p.cluster_max_distance);
        }
        const unsigned long long mask = __ballot(want);
        if (lane_id() == 0) want_mask[k0 >> 6] = mask;
        total += (int)__popcll(mask);
    }
    if (total == 0) return;
    unsigned long long w0 = 0;
    if (lane_id() == 0) w0 = atomicAdd(n_work, (unsigned long long)total);
    w0 = ((unsigned long long)(unsigned)__shfl((int)(w0 >> 32), 0, 64) << 32) | (unsigned)__shfl((int)w0, 0, 64);
    __syncthreads();
    for (int k0 = 0; k0 < npairs; k0 += 64) {
        const unsigned long long mask = want_mask[k0 >> 6];
        const int k = k0 + lane_id();
        if ((mask >> lane_id()) & 1ull) {
            int i = (int)((two_n1 - sqrt(two_n1 * two_n1 - 8.0 * k)) * 0.5);
            if (i < 0) i = 0;
            if (i > ns - 2) i = ns - 2;
            while (i > 0 && i * ns - i * (i + 1) / 2 > k) i--;
            while (i < ns - 2 && (i + 1) * ns - (i + 1) * (i + 2) / 2 <= k) i++;
            const int j = i + 1 + (k - (i * ns - i * (i + 1) / 2));
            const unsigned long long w = w0 + (unsigned long long)__popcll(mask & lanemask_lt());
            if ((long long)w < work_cap) { EditWork e; e.a = m_g[i]; e.b = m_g[j]; e.slot = base + k; work[w] = e; }
        }
        w0 += (unsigned long long)__popcll(mask);
    }
}

// ---- svx_pair_distances: span_position_distance of arbitrary pairs through the very device functions the clustering uses ----------
__device__ __forceinline__ Member member_of(const ClusterIn& in, long long g) {
    Member m; m.start = in.start[g]; m.end = in.end[g]; m.pos2 = in.pos2[g]; m.read = in.read_id[g]; m.aux = in.aux[g]; m.gidx = (int)g; m.c2 = in.contig2[g]; m.pad = 0;
    return m;
}
__global__ void k_pairs_need_edit(long long n_pairs, const int64_t* ia, const int64_t* ib, ClusterIn in, svx_params p, EditWork* work, unsigned long long* n_work) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const long long a = ia[k], b = ib[k];
    if (in.type[a] != SVX_INS || in.type[b] != SVX_INS || !ins_needs_edit(member_of(in, a), member_of(in, b), p)) return;
    const unsigned long long w = atomicAdd(n_work, 1ull);
    EditWork e; e.a = (uint32_t)a; e.b = (uint32_t)b; e.slot = k; work[w] = e;
}
__global__ void k_pairs_distance(long long n_pairs, const int64_t* ia, const int64_t* ib, ClusterIn in, svx_params p, const int32_t* ed, double* out) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    out[k] = span_position_distance(in.type[ia[k]], member_of(in, ia[k]), member_of(in, ib[k]), p, ed[k]);
}
int svx_pair_distances_impl(svx_ctx* c, const ClusterIn& in, int64_t n_pairs, const int64_t* ia_dev, const int64_t* ib_dev, const svx_params* pp, double* out_dev) {
    hipStream_t st = c->stream;
    if (n_pairs <= 0) return SVX_OK;
    SVXCHK(c->counters.reserve(16 * 8));
    unsigned long long* cnt = c->counters.as<unsigned long long>();
    HIPCHK(hipMemsetAsync(cnt, 0, 16 * 8, st));
    SVXCHK(c->work.reserve((size_t)n_pairs * sizeof(EditWork)));
    SVXCHK(c->ed.reserve((size_t)(n_pairs + 1) * 4));
    HIPCHK(hipMemsetAsync(c->ed.p, 0, (size_t)(n_pairs + 1) * 4, st));
    k_pairs_need_edit<<<(unsigned)((n_pairs + 255) / 256), 256, 0, st>>>(n_pairs, ia_dev, ib_dev, in, *pp, c->work.as<EditWork>(), cnt + 8);
    unsigned long long n_work = 0;
    SVXCHK(svx_mail_read(c, st, cnt + 8, 1, &n_work));
    if (n_work) {
        if (!c->g_off_p) return svx_fail(SVX_E_STATE, "svx_set_genome must precede insertion distances", __FILE__, __LINE__, hipSuccess);
        SVXCHK(svx_launch_edit_pairs(c, (int64_t)n_work, c->work.p, in, c->ed.as<int32_t>(), nullptr));
    }
    k_pairs_distance<<<(unsigned)((n_pairs + 255) / 256), 256, 0, st>>>(n_pairs, ia_dev, ib_dev, in, *pp, c->ed.as<int32_t>(), out_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return SVX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// nn-chain average linkage + flat cut, entirely in LDS (scipy _hierarchy.nn_chain / label / cluster_dist)
// ---------------------------------------------------------------------------------------------------------
struct LinkLds {
    double* D;        // condensed distances, n(n-1)/2
    double* Zh;       // merge heights [n-1] in merge order
    double* MD;       // max height below, per row of the height-sorted merge table
    int* Zx; int* Zy; // merge members (slot indices, merge order)
    int* chain;
    int* ord;         // merge order after the stable sort
    int* labels;
};

__device__ __forceinline__ int cidx24(int n, int i, int j) {        // cidx with full-rate 24-bit multiplies (n <= MAXN)
    if (i > j) { const int t = i; i = j; j = t; }
    return (int)__umul24((unsigned)n, (unsigned)i) - (int)(__umul24((unsigned)i, (unsigned)(i + 1)) >> 1) + (j - i - 1);
}

// returns number of flat clusters; labels[0..n) in 1..ncl.  All 64 lanes must call (one wave per problem, n <= 128).
// Everything indexed by a cluster slot or a tree row that the serial parts of scipy's algorithm walk lives in REGISTERS spread over
// the lanes (lane l: entries l and l + 64; lane_table_get = v_readlane with a wave-uniform index), only the distance matrix and
// the merge table are in LDS; control flow is wave-uniform throughout.
__device__ int linkage_fcluster_lds(int n, const LinkLds& w, double cutoff) {
    const int lane = lane_id();
    if (n == 1) { if (lane == 0) w.labels[0] = 1; __syncthreads(); return 1; }
    int sz0 = lane < n ? 1 : 0, sz1 = lane + 64 < n ? 1 : 0;            // cluster sizes per slot, 0 = dead
    int chain_len = 0;
    for (int k = 0; k < n - 1; k++) {
        int x, yprev;
        if (chain_len == 0) {
            // first live cluster
            const unsigned long long b0 = __ballot(sz0 > 0);
            if (b0) x = __ffsll((long long)b0) - 1;
            else { const unsigned long long b1 = __ballot(sz1 > 0); x = 64 + __ffsll((long long)b1) - 1; }
            if (lane == 0) w.chain[0] = x;
            chain_len = 1;
            yprev = -1;
        } else {
            x = __builtin_amdgcn_readfirstlane(w.chain[chain_len - 1]);
            yprev = chain_len > 1 ? __builtin_amdgcn_readfirstlane(w.chain[chain_len - 2]) : -1;
        }
        int y; double cur;
        for (;;) {
            y = yprev;
            cur = yprev >= 0 ? w.D[cidx24(n, x, yprev)] : __builtin_inf();
            // nearest live neighbour of x: strict '<' scanning i upward => lowest index wins ties, previous element kept on ties
            const int i0 = lane, i1 = lane + 64;
            const bool v0 = i0 < n && i0 != x && sz0 > 0, v1 = i1 < n && i1 != x && sz1 > 0;
            const double d0 = v0 ? w.D[cidx24(n, x, i0)] : __builtin_inf();
            const double d1 = v1 ? w.D[cidx24(n, x, i1)] : __builtin_inf();
            const double m = wave_min_f64(d0 < d1 ? d0 : d1);
            if (m < cur) {
                cur = m;
                const unsigned long long b0 = __ballot(v0 && d0 == m);
                if (b0) y = __ffsll((long long)b0) - 1;
                else { const unsigned long long b1 = __ballot(v1 && d1 == m); y = 64 + __ffsll((long long)b1) - 1; }
            }
            if (yprev >= 0 && y == yprev) break;
            if (lane == 0) w.chain[chain_len] = y;
            chain_len++;
            yprev = x; x = y;
        }
        chain_len -= 2;
        if (x > y) { const int t = x; x = y; y = t; }
        const int nx = lane_table_get(sz0, sz1, x), ny = lane_table_get(sz0, sz1, y);
        if (lane == 0) { w.Zx[k] = x; w.Zy[k] = y; w.Zh[k] = cur; }
        lane_table_set(sz0, sz1, x, 0);
        lane_table_set(sz0, sz1, y, nx + ny);
        // Lance-Williams update for average linkage, same expression as scipy
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = lane + 64 * h;
            if (i >= n || i == y || i == x || (h ? sz1 : sz0) == 0) continue;
            const int iy = cidx24(n, i, y);
            w.D[iy] = ((double)nx * w.D[cidx24(n, i, x)] + (double)ny * w.D[iy]) / (double)(nx + ny);
        }
        __syncthreads();
    }
    const int nm = n - 1;
    // stable sort of merges by height: rank = #{h_j < h_i} + #{h_j == h_i, j < i}
    for (int i = lane; i < nm; i += 64) {
        const double h = w.Zh[i]; int r = 0;
        for (int j = 0; j < nm; j++) { const double hj = w.Zh[j]; r += (hj < h) || (hj == h && j < i); }
        w.ord[r] = i;
    }
    __syncthreads();
    // scipy's label(): the union-find over the sorted rows becomes "root node of every leaf" (one register pair), relabelled by all lanes at once;
    // the sorted tree (children, max-height-below <= cutoff) goes into lane tables for the walk below
    int root0 = lane, root1 = lane + 64 < n ? lane + 64 : -1;
    int L0 = 0, L1 = 0, R0 = 0, R1 = 0, le0 = 0, le1 = 0;
    for (int r = 0; r < nm; r++) {
        const int src = __builtin_amdgcn_readfirstlane(w.ord[r]);
        const int a0 = __builtin_amdgcn_readfirstlane(w.Zx[src]), b0 = __builtin_amdgcn_readfirstlane(w.Zy[src]);
        const int a = lane_table_get(root0, root1, a0), b = lane_table_get(root0, root1, b0);
        const int l = a < b ? a : b, rr = a < b ? b : a;
        double m = w.Zh[src];
        if (l >= n) { const double t = w.MD[l - n]; if (t > m) m = t; }
        if (rr >= n) { const double t = w.MD[rr - n]; if (t > m) m = t; }
        if (lane == 0) w.MD[r] = m;
        lane_table_set(L0, L1, r, l);
        lane_table_set(R0, R1, r, rr);
        lane_table_set(le0, le1, r, m <= cutoff ? 1 : 0);
        root0 = (root0 == a || root0 == b) ? n + r : root0;
        root1 = (root1 == a || root1 == b) ? n + r : root1;
    }
    // cluster_monocrit: DFS from the root, left child first (stack, visited flags and labels in lane tables; rows are node id - n)
    int vis0 = 0, vis1 = 0, lab0 = 0, lab1 = 0, stk0 = 0, stk1 = 0;
    int k = 0, ncl = 0, leader = -1;
    lane_table_set(stk0, stk1, 0, nm - 1);
    while (k >= 0) {
        const int root = lane_table_get(stk0, stk1, k);
        const int lc = lane_table_get(L0, L1, root), rc = lane_table_get(R0, R1, root);
        if (leader == -1 && lane_table_get(le0, le1, root)) { leader = root; ncl++; }
        if (lc >= n && !lane_table_get(vis0, vis1, lc - n)) { lane_table_set(vis0, vis1, lc - n, 1); k++; lane_table_set(stk0, stk1, k, lc - n); continue; }
        if (rc >= n && !lane_table_get(vis0, vis1, rc - n)) { lane_table_set(vis0, vis1, rc - n, 1); k++; lane_table_set(stk0, stk1, k, rc - n); continue; }
        if (lc < n) { if (leader == -1) ncl++; lane_table_set(lab0, lab1, lc, ncl); }
        if (rc < n) { if (leader == -1) ncl++; lane_table_set(lab0, lab1, rc, ncl); }
        if (leader == root) leader = -1;
        k--;
    }
    if (lane < n) w.labels[lane] = lab0;
    if (lane + 64 < n) w.labels[lane + 64] = lab1;
    __syncthreads();
    return ncl;
}

__device__ __forceinline__ LinkLds carve_link(char*& sm, int nmax) {
    LinkLds w;
    const int np = nmax * (nmax - 1) / 2;
    w.D = reinterpret_cast<double*>(sm); sm += sizeof(double) * (size_t)(np > 0 ? np : 1);
    w.Zh = reinterpret_cast<double*>(sm); sm += sizeof(double) * nmax;
    w.MD = reinterpret_cast<double*>(sm); sm += sizeof(double) * nmax;
    int* ip = reinterpret_cast<int*>(sm);
    w.Zx = ip; ip += nmax; w.Zy = ip; ip += nmax; w.chain = ip; ip += nmax; w.ord = ip; ip += nmax; w.labels = ip; ip += nmax;
    sm = reinterpret_cast<char*>(ip);
    return w;
}
static size_t link_lds_bytes(int nmax) {
    const int np = nmax * (nmax - 1) / 2;
    return sizeof(double) * (size_t)(np > 0 ? np : 1) + 2 * sizeof(double) * nmax + sizeof(int) * 5 * (size_t)nmax + 16;
}

// utility / test entry: batch of condensed matrices -> flat labels
__global__ __launch_bounds__(64) void k_linkage_batch(long long n_problems, const int32_t* ns, const int64_t* d_off, const double* d, double cutoff,
                                                      const int64_t* label_off, int32_t* labels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long q = blockIdx.x;
    if (q >= n_problems) return;
    const int n = ns[q];
    char* sm = smem;
    LinkLds w = carve_link(sm, MAXN);
    const int np = n * (n - 1) / 2;
    for (int i = lane_id(); i < np; i += 64) w.D[i] = d[d_off[q] + i];
    __syncthreads();
    linkage_fcluster_lds(n, w, cutoff);
    for (int i = lane_id(); i < n; i += 64) labels[label_off[q] + i] = w.labels[i];
}

int svx_linkage_batch(svx_ctx* c, int64_t n_problems, const int32_t* n_dev, const int64_t* d_off_dev, const double* d_dev, double cutoff,
                      const int64_t* label_off_dev, int32_t* labels_dev) {
    if (n_problems <= 0) return SVX_OK;
    const size_t lds = link_lds_bytes(MAXN);
    k_linkage_batch<<<(unsigned)n_problems, 64, lds, c->stream>>>(n_problems, n_dev, d_off_dev, d_dev, cutoff, label_off_dev, labels_dev);
    HIPCHK(hipGetLastError());
    return SVX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// per-partition clustering + consolidation
// ---------------------------------------------------------------------------------------------------------
struct Stage {          // staging records, indexed by samp_base[p] + label-1
    uint8_t* type; uint8_t* aux; int32_t* contig; int32_t* start; int32_t* end; int32_t* contig2; int32_t* start2; int32_t* end2;
    double* score; double* std_span; double* std_pos; int32_t* size; int32_t* mem_local;   // offset of the member list inside the partition
    int32_t* members;   // [samp_total] global signature indices, partition-major then cluster-major
};

__device__ __forceinline__ double stdev_seq(const double* x, int n) {          // FP64 restatement, same op order as the oracle
    double s = 0;
    for (int i = 0; i < n; i++) s += x[i];
    const double c = s / (double)n;
    double ss = 0, sd = 0;
    for (int i = 0; i < n; i++) { const double d = x[i] - c; ss += d * d; sd += d; }
    ss -= sd * sd / (double)n;
    if (ss < 0) ss = 0;
    return sqrt(ss / (double)(n - 1));
}

__device__ __forceinline__ double calc_score(int num, bool has, double std_span, double std_pos, double span) {
    double sds = 0, pds = 0;
    if (has) { const double a = std_span / span; sds = 1 - (a < 1 ? a : 1); const double b = std_pos / span; pds = 1 - (b < 1 ? b : 1); }
    return (double)num + sds * ((double)num / 8) + pds * ((double)num / 8);
}

// consolidate the members with label `lab` (in index order); executed by ONE lane
__device__ void consolidate_one(int t, int contig, const Member* mem, const int* labels, int nm, int lab, double* xs, double* xp,
                                const Stage& st, long long slot, long long mbase, int moff) {
    long long ss = 0, se = 0, ds = 0; int n = 0; long long maxc = 0; int cnt[5] = {0, 0, 0, 0, 0}; int aux0 = 0, contig2 = -1;
    for (int i = 0; i < nm; i++) if (labels[i] == lab) {
        const Member& m = mem[i];
        if (n == 0) { aux0 = m.aux; contig2 = m.c2; }
        ss += m.start; se += m.end; ds += m.pos2; if (m.pos2 > maxc) maxc = m.pos2;
        if (t == SVX_INV && m.aux >= 0 && m.aux < 5) cnt[m.aux]++;
        xs[n] = (double)((long long)m.end - m.start); xp[n] = (double)((long long)m.end + m.start) / 2.0;
        st.members[mbase + moff + n] = m.gidx;
        n++;
    }
    const double avg_s = (double)ss / (double)n, avg_e = (double)se / (double)n;
    const bool has = n > 1;
    double std_span = __builtin_nan(""), std_pos = __builtin_nan("");
    if (has) { std_span = stdev_seq(xs, n); std_pos = stdev_seq(xp, n); }
    const int start = (int)rint(avg_s), end = (int)rint(avg_e);      // round-half-even, like int(round(x))
    int num = n < 80 ? n : 80;
    if (t == SVX_INV) { const int left = cnt[0] + cnt[1], right = cnt[2] + cnt[3]; const int valid = (left < right ? left : right) + cnt[4]; num = valid < 80 ? valid : 80; }
    int c2 = -1, s2 = 0, e2 = 0, aux = 0; double score, o_span = std_span, o_pos = std_pos;
    if (t <= SVX_INV) score = calc_score(num, has, std_span, std_pos, avg_e - avg_s);
    else if (t == SVX_DUP_TAN) {
        score = calc_score(num, has, std_span, std_pos, avg_e - avg_s);
        c2 = contig; s2 = end; e2 = (int)((long long)end + maxc * ((long long)end - start));
    } else if (t == SVX_DUP_INT) {
        const double davg_s = (double)ds / (double)n, davg_e = (double)(ds + (se - ss)) / (double)n;
        const double span = ((avg_e - avg_s) + (davg_e - davg_s)) / 2.0;
        c2 = contig2; s2 = (int)rint(davg_s); e2 = (int)rint(davg_e);
        if (has) {
            // destination span == source span; destination centre = pos + span/2
            int q = 0;
            for (int i = 0; i < nm; i++) if (labels[i] == lab) { const Member& m = mem[i]; xp[q++] = (double)((long long)m.pos2 + ((long long)m.end - m.start) + m.pos2) / 2.0; }
            const double dsp = stdev_seq(xs, n), dpo = stdev_seq(xp, n);
            o_span = (std_span + dsp) / 2.0; o_pos = (std_pos + dpo) / 2.0;
            score = calc_score(num, true, o_span, o_pos, span);
        } else score = calc_score(num, false, 0, 0, span);
    } else {    // BND
        const double davg_s = (double)ds / (double)n, davg_e = (double)(ds + n) / (double)n;
        c2 = contig2; s2 = (int)rint(davg_s); e2 = (int)rint(davg_e); aux = aux0;
        if (has) {
            int q = 0;
            for (int i = 0; i < nm; i++) if (labels[i] == lab) { const Member& m = mem[i]; xp[q++] = (double)((long long)m.pos2 + 1 + m.pos2) / 2.0; }
            const double dpo = stdev_seq(xp, n);
            o_span = std_pos; o_pos = dpo;
            score = calc_score(num, true, std_pos, dpo, 500.0);
        } else score = calc_score(num, false, 0, 0, 500.0);
    }
    st.type[slot] = (uint8_t)t; st.aux[slot] = (uint8_t)aux; st.contig[slot] = contig; st.start[slot] = start; st.end[slot] = end;
    st.contig2[slot] = c2; st.start2[slot] = s2; st.end2[slot] = e2; st.score[slot] = score; st.std_span[slot] = o_span; st.std_pos[slot] = o_pos;
    st.size[slot] = n; st.mem_local[slot] = moff;
}

// CAP = LDS capacity class: partitions with at most CAP sampled members (and more than LO) are handled by this instantiation;
// the small class needs 1/4 of the LDS, so 3-4x more partitions are resident per CU
template <int CAP, int LO>
__global__ __launch_bounds__(64) void k_cluster(long long n_part, const int64_t* part_start, const uint32_t* sidx, const int32_t* sample_idx,
                                                const int64_t* large_excl, const int64_t* samp_base, const int64_t* pair_off, const int32_t* ed,
                                                ClusterIn in, svx_params p, Stage st, int32_t* ncl_out, int32_t* nmem_out,
                                                unsigned long long* n_pairs_stat, int phase) {
    // phase 0: every partition; 1: all but insertions (they need no edit distances and run beside the edit-distance rounds); 2: insertions
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long pt = blockIdx.x;
    if (pt >= n_part) return;
    const int lane = lane_id();
    const long long ps = part_start[pt], size = part_start[pt + 1] - ps;
    const int ns = size > MAXN ? MAXN : (int)size;
    if (ns > CAP || ns <= LO) return;                                               // the other size class owns this partition
    char* sm = smem;
    LinkLds w = carve_link(sm, CAP);
    Member* mem = reinterpret_cast<Member*>(sm); sm += sizeof(Member) * CAP;       // survivors after dedupe
    Member* all = reinterpret_cast<Member*>(sm); sm += sizeof(Member) * CAP;       // the sample
    int* dup = reinterpret_cast<int*>(sm); sm += sizeof(int) * CAP;
    int* orig = reinterpret_cast<int*>(sm); sm += sizeof(int) * CAP;
    int* cl_off = reinterpret_cast<int*>(sm); sm += sizeof(int) * (CAP + 2);
    double* xs = reinterpret_cast<double*>(sm); sm += sizeof(double) * 4;           // scratch of a single-member cluster; larger ones are carved from D below

    const long long sbase = samp_base[pt];
    const uint32_t g0 = member_gidx(ps, size, 0, sidx, sample_idx, large_excl, pt);
    const int t = in.type[g0];
    if ((phase == 1 && t == SVX_INS) || (phase == 2 && t != SVX_INS)) return;
    const int contig = in.contig[g0];
    for (int q = lane; q < ns; q += 64) {
        const uint32_t g = member_gidx(ps, size, q, sidx, sample_idx, large_excl, pt);
        Member m; m.start = in.start[g]; m.end = in.end[g]; m.pos2 = in.pos2[g]; m.read = in.read_id[g]; m.aux = in.aux[g]; m.gidx = (int)g; m.c2 = in.contig2[g]; m.pad = 0;
        all[q] = m; dup[q] = 0;
    }
    __syncthreads();
    const long long pbase = pair_off[pt];
    // same-read duplicates (SVIM_clustering.py:141-151): j is dropped when ANY earlier i of the same read is within the cut
    if (t != SVX_INV && ns > 1) {
        const int npairs = ns * (ns - 1) / 2;
        int i = 0, rem = lane;                                                         // pair k = (i, i + 1 + rem), carried from k to k + 64
        for (int k = lane; k < npairs; k += 64, rem += 64) {
            while (rem >= ns - 1 - i) { rem -= ns - 1 - i; i++; }
            const int j = i + 1 + rem;
            if (all[i].read == all[j].read) {
                const int e = (t == SVX_INS && ins_needs_edit(all[i], all[j], p)) ? ed[pbase + k] : 0;
                if (span_position_distance(t, all[i], all[j], p, e) <= p.cluster_max_distance) dup[j] = 1;
            }
        }
    }
    __syncthreads();
    // order-preserving compaction of the survivors
    int nm;
    {
        const int f0 = (lane < ns && !dup[lane]) ? 1 : 0, f1 = (lane + 64 < ns && !dup[lane + 64]) ? 1 : 0;
        const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
        const int c0 = __popcll(b0);
        if (f0) { const int d = __popcll(b0 & lanemask_lt()); mem[d] = all[lane]; orig[d] = lane; }
        if (f1) { const int d = c0 + __popcll(b1 & lanemask_lt()); mem[d] = all[lane + 64]; orig[d] = lane + 64; }
        nm = c0 + __popcll(b1);
    }
    __syncthreads();
    int ncl;
    if (nm == 1) { if (lane == 0) w.labels[0] = 1; ncl = 1; __syncthreads(); }
    else {
        const int npairs = nm * (nm - 1) / 2;
        int i = 0, rem = lane;
        for (int k = lane; k < npairs; k += 64, rem += 64) {
            while (rem >= nm - 1 - i) { rem -= nm - 1 - i; i++; }
            const int j = i + 1 + rem;
            double d;
            if (t != SVX_INV && mem[i].read == mem[j].read) d = 99999.0;
            else {
                const int e = (t == SVX_INS && ins_needs_edit(mem[i], mem[j], p)) ? ed[pbase + cidx(ns, orig[i], orig[j])] : 0;
                d = span_position_distance(t, mem[i], mem[j], p, e);
            }
            w.D[k] = d;
        }
        if (lane == 0) atomicAdd(n_pairs_stat, (unsigned long long)npairs);
        __syncthreads();
        ncl = linkage_fcluster_lds(nm, w, p.cluster_max_distance);
    }
    // member-list offsets per label (members of a cluster stay in index order)
    if (lane == 0) {
        for (int l = 0; l <= ncl; l++) cl_off[l] = 0;
        for (int i = 0; i < nm; i++) cl_off[w.labels[i]]++;
        int acc = 0;
        for (int l = 1; l <= ncl; l++) { const int c = cl_off[l]; cl_off[l] = acc; acc += c; }
        ncl_out[pt] = ncl; nmem_out[pt] = nm;
    }
    __syncthreads();
    // one lane per cluster; its FP64 scratch (2 x size doubles) is carved from the D region, which is dead now
    double* scratch = w.D;
    for (int l = lane + 1; l <= ncl; l += 64) {
        const int moff = cl_off[l];
        double* mxs = (nm <= 1) ? xs : scratch + 2 * moff;
        const int csz = ((l < ncl) ? cl_off[l + 1] : nm) - moff;
        consolidate_one(t, contig, mem, w.labels, nm, l, mxs, mxs + csz, st, sbase + (l - 1), sbase, moff);
    }
}

// copy staged clusters of partition p to their final dense position
__global__ __launch_bounds__(64) void k_finalize(long long n_part, const int64_t* samp_base, const int32_t* ncl, const int64_t* clu_off,
                                                 const int64_t* mem_off, Stage st, uint8_t* o_type, uint8_t* o_aux, int32_t* o_contig, int32_t* o_start,
                                                 int32_t* o_end, int32_t* o_contig2, int32_t* o_start2, int32_t* o_end2, double* o_score, double* o_span,
                                                 double* o_pos, int32_t* o_size, int64_t* o_moff, int64_t* o_part, uint64_t* o_key, const int32_t* rank) {
    const long long pt = blockIdx.x;
    if (pt >= n_part) return;
    const int n = ncl[pt];
    const long long sb = samp_base[pt], cb = clu_off[pt], mb = mem_off[pt];
    for (int k = lane_id(); k < n; k += 64) {
        const long long s = sb + k, d = cb + k;
        const int t = st.type[s];
        o_type[d] = st.type[s]; o_aux[d] = st.aux[s]; o_contig[d] = st.contig[s]; o_start[d] = st.start[s]; o_end[d] = st.end[s];
        o_contig2[d] = st.contig2[s]; o_start2[d] = st.start2[s]; o_end2[d] = st.end2[s]; o_score[d] = st.score[s]; o_span[d] = st.std_span[s];
        o_pos[d] = st.std_pos[s]; o_size[d] = st.size[s]; o_moff[d] = mb + st.mem_local[s]; o_part[d] = pt;
        // final order: type-major; unilocal types by (contig name rank, start+end) (SVIM_clustering.py:381), bilocal types keep partition order
        uint64_t key = (uint64_t)t << 60;
        if (t <= SVX_INV) key |= ((uint64_t)rank[st.contig[s]] << 34) | (uint64_t)((long long)st.start[s] + st.end[s] + (1ll << 32));
        o_key[d] = key;
    }
}

__global__ void k_copy_members(long long n_part, const int64_t* samp_base, const int32_t* nmem, const int64_t* mem_off, const int32_t* src, int32_t* dst) {
    const long long pt = blockIdx.x;
    if (pt >= n_part) return;
    const int n = nmem[pt];
    for (int k = threadIdx.x; k < n; k += blockDim.x) dst[mem_off[pt] + k] = src[samp_base[pt] + k];
}

// apply the final stable ordering
__global__ void k_permute_clusters(long long n, const uint32_t* perm, const uint8_t* i_type, const uint8_t* i_aux, const int32_t* i_contig,
                                   const int32_t* i_start, const int32_t* i_end, const int32_t* i_contig2, const int32_t* i_start2, const int32_t* i_end2,
                                   const double* i_score, const double* i_span, const double* i_pos, const int32_t* i_size, const int64_t* i_moff,
                                   const int64_t* i_part, uint8_t* o_type, uint8_t* o_aux, int32_t* o_contig, int32_t* o_start, int32_t* o_end,
                                   int32_t* o_contig2, int32_t* o_start2, int32_t* o_end2, double* o_score, double* o_span, double* o_pos, int32_t* o_size,
                                   int64_t* o_src_moff, int64_t* o_part) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { o_size[n] = 0; return; }
    const uint32_t s = perm[i];
    o_type[i] = i_type[s]; o_aux[i] = i_aux[s]; o_contig[i] = i_contig[s]; o_start[i] = i_start[s]; o_end[i] = i_end[s]; o_contig2[i] = i_contig2[s];
    o_start2[i] = i_start2[s]; o_end2[i] = i_end2[s]; o_score[i] = i_score[s]; o_span[i] = i_span[s]; o_pos[i] = i_pos[s]; o_size[i] = i_size[s];
    o_src_moff[i] = i_moff[s]; o_part[i] = i_part[s];
}

__global__ void k_gather_members(long long n, const int32_t* size, const int64_t* src_moff, const int64_t* dst_moff, const int32_t* src, int32_t* dst) {
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int m = size[i];
    for (int k = lane_id(); k < m; k += 64) dst[dst_moff[i] + k] = src[src_moff[i] + k];
}

// clusters are sorted by type: count[t] = lower_bound(t+1) - lower_bound(t)
__global__ void k_type_counts(long long n, const uint8_t* type, unsigned long long* counts) {
    const int t = threadIdx.x;
    if (t >= SVX_NTYPES) return;
    long long lo = 0, hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((int)type[mid] < t) lo = mid + 1; else hi = mid; }
    const long long a = lo;
    hi = n;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((int)type[mid] <= t) lo = mid + 1; else hi = mid; }
    counts[t] = (unsigned long long)(lo - a);
}

// MT19937 state after random.seed(1524): init_by_array([1524]) (constant of the path; generated once on the host)
static void mt_seed_state(uint32_t key0, uint32_t* s) {
    s[0] = 19650218u;
    for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t)i;
    int i = 1;
    for (int k = 624; k; k--) {
        s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1664525u)) + key0 + 0u;
        i++;
        if (i >= 624) { s[0] = s[623]; i = 1; }
    }
    for (int k = 623; k; k--) {
        s[i] = (s[i] ^ ((s[i - 1] ^ (s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s[0] = s[623]; i = 1; }
    }
    s[0] = 0x80000000u;
}

#define GRID(n, t) (unsigned)(((n) + (t) - 1) / (t))

// host-side plan of the consumption tables (k_sample_tables .. k_chase_*)
struct SamplePlan {
    std::vector<int32_t> info;                      // [2 n_large]: type, size of every large partition (sorted order)
    std::vector<SampleMeta> meta; std::vector<ChaseRun> runs;
    long long type_begin[SVX_NTYPES + 1], type_run_begin[SVX_NTYPES + 1];
    long long slots = 0, end_slots = 0, n_runs = 0; int max_width = 0;
    bool ok = false;
    SampleMeta* meta_dev = nullptr; ChaseRun* runs_dev = nullptr; long long* run_start_dev = nullptr; long long* trb_dev = nullptr; long long* ends_dev = nullptr;
};

// words one pool-method sample of a partition of n members (101..1045) consumes: draw i accepts a word with probability (n-i) / 2^k
static void sample_moments(int n_q, double* mean, double* var) {
    static thread_local double moments_mean[1046], moments_var[1046];
    static thread_local bool moments_have[1046];
    if (!moments_have[n_q]) {
        double dm = 0, dv = 0;
        for (int i = 0; i < 100; i++) {
            const double bound = (double)(n_q - i);
            int k = 0; while ((1u << k) <= (unsigned)(n_q - i)) k++;
            const double pacc = bound / (double)(1u << k);
            dm += 1.0 / pacc; dv += (1.0 - pacc) / (pacc * pacc);
        }
        moments_mean[n_q] = dm; moments_var[n_q] = dv; moments_have[n_q] = true;
    }
    *mean = moments_mean[n_q]; *var = moments_var[n_q];
}

// windows (expected start -+ 6 sigma) of every large partition, given mean / variance / lower bound of each type's first start
static void plan_sample_tables(SamplePlan& P, long long n_large, const double* mean0, const double* var0, const long long* least0) {
    P.meta.assign((size_t)n_large, SampleMeta{0, 0, 0, 0});
    P.runs.clear();
    P.slots = 0; P.end_slots = 0; P.max_width = 0; P.ok = true;
    const std::vector<int32_t>& info = P.info;
    long long q = 0;
    for (int t = 0; t <= SVX_NTYPES; t++) P.type_begin[t] = n_large;
    for (int t = 0; t < SVX_NTYPES && P.ok; t++) {
        while (q < n_large && info[(size_t)q * 2] < t) q++;
        P.type_begin[t] = q;
        double mean = mean0[t], var = var0[t];
        long long least = least0[t];
        for (; q < n_large && info[(size_t)q * 2] == t; q++) {
            const int n_q = info[(size_t)q * 2 + 1];
            if (n_q > 1045) { P.ok = false; break; }                       // set method: the walk depends on the values drawn
            const double sd = sqrt(var);
            long long lo = (long long)floor(mean - 6.0 * sd) - 32, hi = (long long)ceil(mean + 6.0 * sd) + 32;
            if (lo < least) lo = least;
            SampleMeta& m = P.meta[(size_t)q];
            m.lo = lo; m.width = (int)(hi - lo + 1); m.n = n_q; m.off = P.slots;
            P.slots += m.width;
            if (m.width > P.max_width) P.max_width = m.width;
            double dm, dv;
            sample_moments(n_q, &dm, &dv);
            mean += dm; var += dv;
            least += 100;
        }
    }
    P.type_begin[SVX_NTYPES] = n_large;
    for (int t = SVX_NTYPES - 1; t >= 0; t--) if (P.type_begin[t] > P.type_begin[t + 1]) P.type_begin[t] = P.type_begin[t + 1];
    if (!(P.ok && P.slots > 0 && P.slots < (1ll << 31) && n_large <= 65535)) { P.ok = false; return; }
    // runs of CHASE_RUN partitions, never across a type boundary
    for (int t = 0; t < SVX_NTYPES; t++) {
        P.type_run_begin[t] = (long long)P.runs.size();
        for (long long f = P.type_begin[t]; f < P.type_begin[t + 1]; f += CHASE_RUN) {
            ChaseRun r; r.first = f; r.last = f + CHASE_RUN < P.type_begin[t + 1] ? f + CHASE_RUN : P.type_begin[t + 1];
            r.eoff = P.end_slots; r.type = t; r.pad = 0;
            P.end_slots += P.meta[(size_t)f].width;
            P.runs.push_back(r);
        }
    }
    P.type_run_begin[SVX_NTYPES] = (long long)P.runs.size();
    P.n_runs = (long long)P.runs.size();
}

// transfer table of one rank: for every candidate start of each type's first window, the stream position after ALL of this rank's large
// partitions of that type (-1: the walk left a window)
__global__ __launch_bounds__(256) void k_chase_span(const SampleMeta* meta, const ChaseRun* runs, const long long* type_run_begin, const long long* ends,
                                                    const long long* f_off /* [NTYPES + 1] */, long long* f) {
    const int t = blockIdx.y;
    const long long b0 = type_run_begin[t], b1 = type_run_begin[t + 1];
    if (b0 >= b1) return;
    const SampleMeta m0 = meta[runs[b0].first];
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= m0.width) return;
    long long pos = m0.lo + s;
    for (long long b = b0; b < b1; b++) {
        const ChaseRun r = runs[b];
        const SampleMeta m = meta[r.first];
        const long long idx = pos - m.lo;
        if (idx < 0 || idx >= m.width) { pos = -1; break; }
        pos = ends[r.eoff + idx];
        if (pos < 0) break;
    }
    f[f_off[t] + s] = pos;
}

// ---- contig-sharded ranks: where do this rank's sample streams start? ------------------------------------------------------------------------
// All communication is an all-gather of small host buffers through the callback of svx_cluster_set_ranks; every logical message is a 128-byte header
// (status, payload size, inline values) followed by an optional payload padded to the largest one.  A rank that fails sends a header with a
// negative status instead of its next header, so the others fail too instead of waiting for it.
struct XHdr { int64_t status, payload, a[14]; };

static int xchg_headers(svx_ctx* c, const XHdr& mine, std::vector<XHdr>& all) {
    all.assign((size_t)c->xr_world, XHdr{});
    if (c->xr_fn(c->xr_user, &mine, all.data(), (int64_t)sizeof(XHdr)) != 0) { c->xr_pending = false; return svx_fail(SVX_E_STATE, "rank exchange: the all-gather callback failed", __FILE__, __LINE__, hipSuccess); }
    for (int r = 0; r < c->xr_world; r++)
        if (all[(size_t)r].status < 0) { c->xr_pending = false; return svx_fail(SVX_E_STATE, "rank exchange: another rank reported a failure", __FILE__, __LINE__, hipSuccess); }
    return SVX_OK;
}
static int xchg_payload(svx_ctx* c, const void* mine, size_t my_bytes, size_t each, std::vector<char>& all) {
    std::vector<char> send(each, 0);
    if (my_bytes) memcpy(send.data(), mine, my_bytes);
    all.assign(each * (size_t)c->xr_world, 0);
    if (c->xr_fn(c->xr_user, send.data(), all.data(), (int64_t)each) != 0) { c->xr_pending = false; return svx_fail(SVX_E_STATE, "rank exchange: the all-gather callback failed", __FILE__, __LINE__, hipSuccess); }
    return SVX_OK;
}
// the other ranks are (or will be) waiting in a header exchange: tell them this rank gave up
void svx_exchange_poison(svx_ctx* c) {
    if (!c->xr_pending || c->xr_world <= 1 || !c->xr_fn) return;
    c->xr_pending = false;
    XHdr h{}; h.status = -1;
    std::vector<XHdr> all((size_t)c->xr_world);
    (void)c->xr_fn(c->xr_user, &h, all.data(), (int64_t)sizeof(XHdr));
}

template <class SpecTables, class SpecFinish, class ExactFrom>
static int svx_sampling_exchange(svx_ctx* c, long long n_large, const int32_t* info, SpecTables spec_tables, SpecFinish spec_finish, ExactFrom exact_from) {
    const int W = c->xr_world, R = c->xr_rank;
    // (1) sizes of everybody's large partitions, per type
    XHdr h{}; h.status = 0; h.payload = n_large * 4; h.a[0] = n_large;
    std::vector<int32_t> mine((size_t)n_large + 1);
    for (long long q = 0; q < n_large; q++) { h.a[1 + info[2 * q]]++; mine[(size_t)q] = info[2 * q + 1]; }
    std::vector<XHdr> H;
    SVXCHK(xchg_headers(c, h, H));
    size_t each = 0;
    for (auto& x : H) each = (size_t)x.payload > each ? (size_t)x.payload : each;
    std::vector<char> sizes_all;
    if (each) SVXCHK(xchg_payload(c, mine.data(), (size_t)n_large * 4, each, sizes_all));
    bool spec = true;
    double mean0[SVX_NTYPES], var0[SVX_NTYPES]; long long least0[SVX_NTYPES];
    for (int t = 0; t < SVX_NTYPES; t++) { mean0[t] = 0; var0[t] = 0; least0[t] = 0; }
    for (int r = 0; r < W; r++) {
        const int32_t* sz = each ? reinterpret_cast<const int32_t*>(sizes_all.data() + each * (size_t)r) : nullptr;
        if (H[(size_t)r].a[0] > 65535) spec = false;
        long long at = 0;
        for (int t = 0; t < SVX_NTYPES; t++)
            for (long long k = 0; k < H[(size_t)r].a[1 + t]; k++, at++) {
                const int n_q = sz[at];
                if (n_q > 1045) { spec = false; continue; }                 // set method somewhere: its consumption depends on the values drawn
                if (r < R) { double dm, dv; sample_moments(n_q, &dm, &dv); mean0[t] += dm; var0[t] += dv; least0[t] += 100; }
            }
    }
    if (getenv("SVX_RANKS_EXACT")) spec = false;              // test hook (set for every rank alike): always the exact rounds
    long long start[SVX_NTYPES] = {0, 0, 0, 0, 0, 0};
    if (spec) {
        // (2) every rank's transfer tables around its expected starts, built concurrently
        long long lo[SVX_NTYPES], width[SVX_NTYPES]; const long long* f = nullptr; bool ok = true;
        SVXCHK(spec_tables(mean0, var0, least0, lo, width, &f, &ok));
        XHdr h2{}; h2.status = 0; h2.a[0] = ok ? 0 : 1;
        long long tot = 0;
        for (int t = 0; t < SVX_NTYPES; t++) { h2.a[1 + t] = lo[t]; h2.a[7 + t] = ok ? width[t] : 0; tot += ok ? width[t] : 0; }
        h2.payload = tot * 8;
        SVXCHK(xchg_headers(c, h2, H));
        each = 0;
        for (auto& x : H) { each = (size_t)x.payload > each ? (size_t)x.payload : each; if (x.a[0]) spec = false; }
        std::vector<char> f_all;
        if (each) SVXCHK(xchg_payload(c, f, (size_t)tot * 8, each, f_all));
        if (spec) {
            // (3) compose: rank r starts where the chain through ranks 0..r-1 ends (every rank computes every rank's start: the same verdict everywhere)
            long long pos[SVX_NTYPES] = {0, 0, 0, 0, 0, 0};
            for (int r = 0; r < W && spec; r++) {
                if (r == R) for (int t = 0; t < SVX_NTYPES; t++) start[t] = pos[t];
                const long long* fr = each ? reinterpret_cast<const long long*>(f_all.data() + each * (size_t)r) : nullptr;
                long long off = 0;
                for (int t = 0; t < SVX_NTYPES; t++) {
                    const long long w = H[(size_t)r].a[7 + t];
                    if (H[(size_t)r].a[0] == 0 && w > 0) {
                        const long long idx = pos[t] - H[(size_t)r].a[1 + t];
                        if (idx < 0 || idx >= w || fr[off + idx] < 0) { spec = false; break; }
                        pos[t] = fr[off + idx];
                    }
                    off += w;
                }
            }
        }
        if (spec) { c->xr_pending = false; return spec_finish(start); }
    }
    // (4) the exact chain, rank after rank (set-method partitions, > 65535 large partitions on a rank, or a start outside its 6-sigma window):
    // round j publishes rank j's end positions
    long long cur[SVX_NTYPES] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < W; j++) {
        XHdr h5{}; h5.status = 0;
        if (j == R) {
            long long end[SVX_NTYPES];
            SVXCHK(exact_from(cur, end));
            for (int t = 0; t < SVX_NTYPES; t++) h5.a[t] = end[t];
        }
        SVXCHK(xchg_headers(c, h5, H));
        for (int t = 0; t < SVX_NTYPES; t++) cur[t] = H[(size_t)j].a[t];
    }
    c->xr_pending = false;
    return SVX_OK;
}

static int svx_cluster_body(svx_ctx* c, const ClusterIn& in, int32_t n_contig, const int32_t* rank, const svx_params* pp) {
    hipStream_t st = c->stream;
    const svx_params p = *pp;
    const int64_t n = in.n;
    c->last_cluster_source_n = n;
    DevClusters& out = c->clu;
    out.n = 0; out.n_members = 0;
    for (int t = 0; t < SVX_NTYPES; t++) out.type_count[t] = 0;
    svx_stats& S = c->stats;
    S.n_partitions = S.n_large_partitions = S.n_pairs = S.n_edit_pairs = S.n_edit_cells = S.n_clusters = S.n_hap_bytes = 0;
    S.n_edit_wordcols_issued = S.n_edit_wordcols_useful = S.n_edit_wordcols_retry = S.n_edit_wordcols_band = 0; S.edit_guess = 0;
    S.t_cluster_ms = S.t_partition_ms = S.t_edit_ms = S.t_linkage_ms = 0;
    for (int t = 0; t < SVX_NTYPES; t++) c->stream_start[t] = c->stream_end[t] = 0;
    if (n == 0) {
        if (c->xr_world > 1 && c->xr_fn) {                  // a rank without signatures still takes part in the exchange (its streams pass through)
            long long pass[SVX_NTYPES] = {0, 0, 0, 0, 0, 0};
            auto spec_tables = [&](const double*, const double*, const long long*, long long* lo, long long* width, const long long** f, bool* ok) -> int {
                for (int t = 0; t < SVX_NTYPES; t++) { lo[t] = 0; width[t] = 0; }
                *f = nullptr; *ok = true;
                return SVX_OK;
            };
            auto spec_finish = [&](const long long* start) -> int { for (int t = 0; t < SVX_NTYPES; t++) pass[t] = start[t]; return SVX_OK; };
            auto exact_from = [&](const long long* start, long long* end) -> int { for (int t = 0; t < SVX_NTYPES; t++) pass[t] = end[t] = start[t]; return SVX_OK; };
            SVXCHK(svx_sampling_exchange(c, 0, nullptr, spec_tables, spec_finish, exact_from));
            for (int t = 0; t < SVX_NTYPES; t++) c->stream_start[t] = c->stream_end[t] = pass[t];
        }
        return SVX_OK;
    }
    if (n >= (1ll << 31)) return svx_fail(SVX_E_ARG, "more than 2^31 signatures in one call", __FILE__, __LINE__, hipSuccess);
    const int T = 256;
    HIPCHK(hipEventRecord(c->ev[8], st));
    SVXCHK(svx_edit_prepack_begin(c, in, p, c->ev[8]));     // haplotype store of the insertions, on a side stream beside everything up to the pair list
    // ---- sort by (type, contig ranks, coordinate), stable w.r.t. list order --------------------------------------
    SVXCHK(c->k_hi.reserve((size_t)n * 8)); SVXCHK(c->k_lo.reserve((size_t)n * 8)); SVXCHK(c->k_idx.reserve((size_t)n * 4));
    SVXCHK(c->k_hi2.reserve((size_t)n * 8)); SVXCHK(c->k_lo2.reserve((size_t)n * 8)); SVXCHK(c->k_idx2.reserve((size_t)n * 4));
    k_make_keys<<<GRID(n, T), T, 0, st>>>(in, rank, c->k_hi.as<uint64_t>(), c->k_lo.as<uint64_t>(), c->k_idx.as<uint32_t>());
    SVXCHK(svx_sort_pairs_u64(c, c->k_lo.as<uint64_t>(), c->k_lo2.as<uint64_t>(), c->k_idx.as<uint32_t>(), c->k_idx2.as<uint32_t>(), n, 0, 32));
    k_gather_u64<<<GRID(n, T), T, 0, st>>>(c->k_hi.as<uint64_t>(), c->k_idx2.as<uint32_t>(), c->k_hi2.as<uint64_t>(), n);
    SVXCHK(svx_sort_pairs_u64(c, c->k_hi2.as<uint64_t>(), c->k_hi.as<uint64_t>(), c->k_idx2.as<uint32_t>(), c->k_idx.as<uint32_t>(), n, 0, 64));
    const uint64_t* hi_sorted = c->k_hi.as<uint64_t>();
    const uint32_t* sidx = c->k_idx.as<uint32_t>();
    // ---- partitions ------------------------------------------------------------------------------------------------
    SVXCHK(c->part_flag.reserve((size_t)(n + 1) * 8)); SVXCHK(c->part_id.reserve((size_t)(n + 1) * 8));
    k_part_flags<<<GRID(n + 1, T), T, 0, st>>>(in, hi_sorted, sidx, p.partition_max_distance, c->part_flag.as<int64_t>());
    SVXCHK(svx_exclusive_scan_i64(c, c->part_flag.as<int64_t>(), c->part_id.as<int64_t>(), n + 1));
    int64_t n_part = 0;
    SVXCHK(svx_mail_read(c, st, c->part_id.as<int64_t>() + n, 1, &n_part));
    SVXCHK(svx_edit_prepack_pack(c, in));
    SVXCHK(c->part_start.reserve((size_t)(n_part + 1) * 8));
    k_part_starts<<<GRID(n + 1, T), T, 0, st>>>(c->part_flag.as<int64_t>(), c->part_id.as<int64_t>(), n, c->part_start.as<int64_t>(), n_part);
    // per-partition sizes -> sample base, large-partition slots, INS pair slots (5 arrays of n_part+1 int64 in part_meta)
    const size_t PM = (size_t)(n_part + 1);
    SVXCHK(c->part_meta.reserve(PM * 8 * 8));
    int64_t* ns_a = c->part_meta.as<int64_t>(); int64_t* large_a = ns_a + PM; int64_t* pairs_a = large_a + PM;
    int64_t* samp_base = pairs_a + PM; int64_t* large_excl = samp_base + PM; int64_t* pair_off = large_excl + PM;
    int64_t* clu_off = pair_off + PM; int64_t* mem_off = clu_off + PM;
    k_part_sizes<<<GRID(n_part + 1, T), T, 0, st>>>(c->part_start.as<int64_t>(), n_part, in.type, sidx, ns_a, large_a, pairs_a);
    SVXCHK(svx_exclusive_scan_i64(c, ns_a, samp_base, n_part + 1));
    SVXCHK(svx_exclusive_scan_i64(c, large_a, large_excl, n_part + 1));
    SVXCHK(svx_exclusive_scan_i64(c, pairs_a, pair_off, n_part + 1));
    int64_t totals[3];
    SVXCHK(svx_mail_read3(c, st, samp_base + n_part, 1, &totals[0], large_excl + n_part, 1, &totals[1], pair_off + n_part, 1, &totals[2]));
    const int64_t samp_total = totals[0], n_large = totals[1], pair_total = totals[2];
    SVXCHK(c->counters.reserve(16 * 8));
    unsigned long long* cnt = c->counters.as<unsigned long long>();
    HIPCHK(hipMemsetAsync(cnt, 0, 16 * 8, st));
    // ---- sampling ----------------------------------------------------------------------------------------------------
    // The word stream of a signature type is consumed by its > 100-member partitions in global order without re-seeding
    // (src/svim/SVIM_clustering.py:129-134).  Single rank: every stream starts at 0.  Contig-sharded ranks (svx_cluster_set_ranks): the global
    // order is rank-major, so a rank's streams continue where the partitions of the ranks before it stop - sampling_exchange() finds those
    // positions with all-gathers only (no rank waits for another rank's sampling).
    long long chain[2 * SVX_NTYPES];
    for (int t = 0; t < 2 * SVX_NTYPES; t++) chain[t] = 0;
    SVXCHK(c->samp_idx.reserve((size_t)(n_large + 1) * 100 * 4));
    SVXCHK(c->samp_chain.reserve(2 * SVX_NTYPES * 8));
    long long* chain_dev = c->samp_chain.as<long long>();
    int* err = reinterpret_cast<int*>(cnt + 15);
    long long* samp_start = nullptr;
    SamplePlan plan;
    auto ensure_stream = [&](long long want) -> int {
        if (c->mt_have >= want) return SVX_OK;
        // (re)generate the prefix of the seed(1524) word stream this context keeps
        const long long blocks = (2 * want + 623) / 624;
        SVXCHK(c->mt_words.reserve((size_t)blocks * 624 * 4 + 624 * 4 + 64));
        uint32_t* mt_dev = c->mt_words.as<uint32_t>() + blocks * 624;
        uint32_t mt_host[624];
        mt_seed_state(1524u, mt_host);
        HIPCHK(hipMemcpyAsync(mt_dev, mt_host, sizeof mt_host, hipMemcpyHostToDevice, st));
        k_mt_generate<<<1, 64, 0, st>>>(mt_dev, c->mt_words.as<uint32_t>(), blocks);
        HIPCHK(hipStreamSynchronize(st));       // mt_host is a stack buffer
        c->mt_have = blocks * 624;
        return SVX_OK;
    };
    if (n_large > 0) {
        SVXCHK(c->large_list.reserve((size_t)n_large * 4 + 64));
        k_large_list<<<GRID(n_part, T), T, 0, st>>>(large_a, large_excl, n_part, c->large_list.as<int32_t>());
        SVXCHK(c->samp_stream.reserve((size_t)n_large * 8 + 64));
        samp_start = c->samp_stream.as<long long>();
        SVXCHK(c->samp_meta.reserve((size_t)n_large * (8 + sizeof(SampleMeta)) + (SVX_NTYPES + 1) * 8 + 64));
        int32_t* info_dev = c->samp_meta.as<int32_t>();
        k_large_info<<<GRID(n_large, T), T, 0, st>>>(c->large_list.as<int32_t>(), n_large, c->part_start.as<int64_t>(), sidx, in.type, info_dev);
        plan.info.resize((size_t)n_large * 2);
        SVXCHK(svx_d2h(plan.info.data(), info_dev, (size_t)n_large * 8, st));
    }
    // consumption tables for windows around the given expected start of every type (mean0 / var0; least0 = lower bound), launched up to k_chase_runs
    auto build_tables = [&](const double* mean0, const double* var0, const long long* least0, long long reach) -> int {
        plan_sample_tables(plan, n_large, mean0, var0, least0);
        if (!plan.ok) return SVX_OK;
        SVXCHK(ensure_stream(reach + n_large * 512 + 4 * 624));           // expected use: <= ~200 words per partition
        SVXCHK(c->samp_table.reserve((size_t)plan.slots * 2 + 64));
        SVXCHK(c->samp_runs.reserve((size_t)plan.n_runs * (sizeof(ChaseRun) + 8) + (size_t)plan.end_slots * 8 + (SVX_NTYPES + 1) * 8 + 64));
        plan.meta_dev = reinterpret_cast<SampleMeta*>(c->samp_meta.as<char>() + (size_t)n_large * 8);
        plan.runs_dev = c->samp_runs.as<ChaseRun>();
        plan.run_start_dev = reinterpret_cast<long long*>(plan.runs_dev + plan.n_runs);
        plan.trb_dev = plan.run_start_dev + plan.n_runs;
        plan.ends_dev = plan.trb_dev + (SVX_NTYPES + 1);
        SVXCHK(svx_h2d(plan.meta_dev, plan.meta.data(), (size_t)n_large * sizeof(SampleMeta), st));
        SVXCHK(svx_h2d(plan.runs_dev, plan.runs.data(), (size_t)plan.n_runs * sizeof(ChaseRun), st));
        HIPCHK(hipMemcpyAsync(plan.trb_dev, plan.type_run_begin, sizeof plan.type_run_begin, hipMemcpyHostToDevice, st));
        k_sample_tables<<<dim3((unsigned)((plan.max_width + 255) / 256), (unsigned)n_large), 256, 0, st>>>(plan.meta_dev, c->mt_words.as<uint32_t>(), c->mt_have,
                                                                                                         c->samp_table.as<uint16_t>());
        k_chase_runs<<<dim3((unsigned)((plan.max_width + 255) / 256), (unsigned)plan.n_runs), 256, 0, st>>>(plan.meta_dev, plan.runs_dev, c->samp_table.as<uint16_t>(), plan.ends_dev);
        HIPCHK(hipGetLastError());
        return SVX_OK;
    };
    // with the tables built: follow the chain from the exact starts in chain[0..5], sample; *done = false when a start left its window
    auto finish_tables = [&](bool* done) -> int {
        HIPCHK(hipMemcpyAsync(chain_dev, chain, sizeof chain, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(err, 0, 8, st));
        k_chase_top<<<SVX_NTYPES, 64, 0, st>>>(plan.meta_dev, plan.runs_dev, plan.trb_dev, plan.ends_dev, plan.run_start_dev, err, chain_dev);
        k_chase_fill<<<GRID(plan.n_runs, 64), 64, 0, st>>>(plan.meta_dev, plan.runs_dev, plan.n_runs, c->samp_table.as<uint16_t>(), plan.run_start_dev, samp_start);
        k_sample_apply<<<(unsigned)n_large, 64, 0, st>>>(c->large_list.as<int32_t>(), n_large, c->part_start.as<int64_t>(), sidx, in.type, large_excl,
                                                        c->mt_words.as<uint32_t>(), c->mt_have, samp_start, c->samp_idx.as<int32_t>());
        HIPCHK(hipGetLastError());
        unsigned long long err_word = 0;              // err is the low half of cnt[15]
        SVXCHK(svx_mail_read2(c, st, cnt + 15, 1, &err_word, chain_dev, (int)(sizeof chain / 8), chain));       // the wait also covers the plan's host vectors
        const int h_err = (int)(uint32_t)err_word;
        *done = !h_err;
        return SVX_OK;
    };
    // this rank's sampling from the exact stream positions chain[0..5]; leaves the end positions in chain[6..11]
    auto exact_sampling = [&]() -> int {
        for (int t = 0; t < SVX_NTYPES; t++) chain[SVX_NTYPES + t] = chain[t];
        if (n_large == 0) return SVX_OK;
        long long chain_max = 0;
        for (int t = 0; t < SVX_NTYPES; t++) chain_max = chain[t] > chain_max ? chain[t] : chain_max;
        long long cap = chain_max + n_large * 512 + 4 * 624;
        bool done = false;
        {
            double mean0[SVX_NTYPES], var0[SVX_NTYPES]; long long least0[SVX_NTYPES];
            for (int t = 0; t < SVX_NTYPES; t++) { mean0[t] = (double)chain[t]; var0[t] = 0; least0[t] = chain[t]; }
            SVXCHK(build_tables(mean0, var0, least0, chain_max));
            if (plan.ok) SVXCHK(finish_tables(&done));
        }
        for (int attempt = 0; !done && attempt < 6; attempt++) {
            SVXCHK(ensure_stream(cap));
            const uint32_t* stream = c->mt_words.as<uint32_t>();
            HIPCHK(hipMemcpyAsync(chain_dev, chain, sizeof chain, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(err, 0, 8, st));
            k_sample_scan<<<SVX_NTYPES, 64, 0, st>>>(c->large_list.as<int32_t>(), n_large, c->part_start.as<int64_t>(), sidx, in.type, large_excl,
                                                    stream, c->mt_have, samp_start, c->samp_idx.as<int32_t>(), err, chain_dev);
            k_sample_apply<<<(unsigned)n_large, 64, 0, st>>>(c->large_list.as<int32_t>(), n_large, c->part_start.as<int64_t>(), sidx, in.type, large_excl,
                                                            stream, c->mt_have, samp_start, c->samp_idx.as<int32_t>());
            HIPCHK(hipGetLastError());
            unsigned long long err_word = 0;
            SVXCHK(svx_mail_read2(c, st, cnt + 15, 1, &err_word, chain_dev + SVX_NTYPES, SVX_NTYPES, chain + SVX_NTYPES));
            const int h_err = (int)(uint32_t)err_word;
            if (!h_err) break;
            if (attempt == 5) return svx_fail(SVX_E_CAPACITY, "random word stream", __FILE__, __LINE__, hipSuccess);
            cap = c->mt_have * 4;
        }
        return SVX_OK;
    };
    if (c->xr_world <= 1 || !c->xr_fn) {
        SVXCHK(exact_sampling());
    } else {
        // speculation: tables around the EXPECTED starts (from the sizes of the earlier ranks' partitions); F = end position of this rank's chain for
        // every candidate start of each type's first window
        std::vector<long long> f_host;
        auto spec_tables = [&](const double* mean0, const double* var0, const long long* least0, long long* lo, long long* width, const long long** f, bool* ok) -> int {
            for (int t = 0; t < SVX_NTYPES; t++) { lo[t] = 0; width[t] = 0; }
            *ok = true; *f = nullptr;
            if (n_large == 0) return SVX_OK;
            double mx = 0;
            for (int t = 0; t < SVX_NTYPES; t++) mx = mean0[t] + 6.0 * sqrt(var0[t]) > mx ? mean0[t] + 6.0 * sqrt(var0[t]) : mx;
            SVXCHK(build_tables(mean0, var0, least0, (long long)mx + 64));
            if (!plan.ok) { *ok = false; return SVX_OK; }
            long long f_off[SVX_NTYPES + 1]; f_off[0] = 0;
            int wmax = 1;
            for (int t = 0; t < SVX_NTYPES; t++) {
                const bool has = plan.type_begin[t] < plan.type_begin[t + 1];
                if (has) { const SampleMeta& m0 = plan.meta[(size_t)plan.type_begin[t]]; lo[t] = m0.lo; width[t] = m0.width; if (m0.width > wmax) wmax = m0.width; }
                f_off[t + 1] = f_off[t] + width[t];
            }
            SVXCHK(c->tmp5.reserve((size_t)(f_off[SVX_NTYPES] + SVX_NTYPES + 2) * 8));
            long long* f_dev = c->tmp5.as<long long>(); long long* foff_dev = f_dev + f_off[SVX_NTYPES];
            HIPCHK(hipMemcpyAsync(foff_dev, f_off, sizeof f_off, hipMemcpyHostToDevice, st));
            k_chase_span<<<dim3((unsigned)((wmax + 255) / 256), SVX_NTYPES), 256, 0, st>>>(plan.meta_dev, plan.runs_dev, plan.trb_dev, plan.ends_dev, foff_dev, f_dev);
            HIPCHK(hipGetLastError());
            f_host.resize((size_t)f_off[SVX_NTYPES] + 1);
            SVXCHK(svx_d2h(f_host.data(), f_dev, (size_t)f_off[SVX_NTYPES] * 8, st));
            *f = f_host.data();
            return SVX_OK;
        };
        auto spec_finish = [&](const long long* start) -> int {
            for (int t = 0; t < SVX_NTYPES; t++) chain[t] = chain[SVX_NTYPES + t] = start[t];
            if (n_large == 0) return SVX_OK;
            bool done = false;
            SVXCHK(finish_tables(&done));
            if (!done) return svx_fail(SVX_E_STATE, "stream positions left the windows their own transfer table covered", __FILE__, __LINE__, hipSuccess);
            return SVX_OK;
        };
        auto exact_from = [&](const long long* start, long long* end) -> int {
            for (int t = 0; t < SVX_NTYPES; t++) chain[t] = start[t];
            SVXCHK(exact_sampling());
            for (int t = 0; t < SVX_NTYPES; t++) end[t] = chain[SVX_NTYPES + t];
            return SVX_OK;
        };
        SVXCHK(svx_sampling_exchange(c, n_large, plan.info.data(), spec_tables, spec_finish, exact_from));
    }
    for (int t = 0; t < SVX_NTYPES; t++) { c->stream_start[t] = chain[t]; c->stream_end[t] = chain[SVX_NTYPES + t]; }
    HIPCHK(hipEventRecord(c->ev[9], st));
    // ---- staging area of the per-partition clustering --------------------------------------------------------------
    const size_t SN = (size_t)(samp_total + 1);
    SVXCHK(c->stage.reserve(SN * (2 + 4 * 7 + 8 * 3) + 64));
    Stage stg;
    {
        char* b = c->stage.as<char>();
        stg.score = reinterpret_cast<double*>(b); b += SN * 8; stg.std_span = reinterpret_cast<double*>(b); b += SN * 8;
        stg.std_pos = reinterpret_cast<double*>(b); b += SN * 8;
        stg.contig = reinterpret_cast<int32_t*>(b); b += SN * 4; stg.start = reinterpret_cast<int32_t*>(b); b += SN * 4;
        stg.end = reinterpret_cast<int32_t*>(b); b += SN * 4; stg.contig2 = reinterpret_cast<int32_t*>(b); b += SN * 4;
        stg.start2 = reinterpret_cast<int32_t*>(b); b += SN * 4; stg.end2 = reinterpret_cast<int32_t*>(b); b += SN * 4;
        stg.size = reinterpret_cast<int32_t*>(b); b += SN * 4;
        stg.type = reinterpret_cast<uint8_t*>(b); b += SN; stg.aux = reinterpret_cast<uint8_t*>(b); b += SN;
    }
    SVXCHK(c->stage_members.reserve(SN * 4 * 2));
    stg.members = c->stage_members.as<int32_t>(); stg.mem_local = stg.members + SN;
    SVXCHK(c->labels.reserve(PM * 4 * 2));
    int32_t* ncl_a = c->labels.as<int32_t>(); int32_t* nmem_a = ncl_a + PM;
    HIPCHK(hipMemsetAsync(ncl_a, 0, PM * 8, st));
    // three LDS size classes: 14 KB (<= 48 members: 11 partitions resident per CU), 30 KB (<= 72: 5), 53 KB (<= 100: 3) - the kernel is
    // bound by the latency of its LDS round trips, so what counts is how many partitions a CU works on at once
    constexpr int SMALL = 48, MID = 72;
    auto cluster_lds = [](int cap) { return link_lds_bytes(cap) + sizeof(Member) * cap * 2 + sizeof(int) * (3 * cap + 2) + sizeof(double) * 4 + 64; };
    // The three size classes are independent: on the main stream (phase 0 / 2, nothing else is running any more) they go to three
    // streams so that their tails overlap; the side-stream pass (phase 1, beside the edit-distance rounds) stays on its one stream.
    auto launch_cluster = [&](hipStream_t ks, int phase) -> int {
        hipStream_t s_mid = ks, s_small = ks;
        const bool fork = phase != 1;
        if (fork) {
            HIPCHK(hipEventRecord(c->ev[15], ks));
            HIPCHK(hipStreamWaitEvent(c->aux[0], c->ev[15], 0)); HIPCHK(hipStreamWaitEvent(c->aux[1], c->ev[15], 0));
            s_mid = c->aux[0]; s_small = c->aux[1];
        }
        k_cluster<MAXN, MID><<<(unsigned)n_part, 64, cluster_lds(MAXN), ks>>>(n_part, c->part_start.as<int64_t>(), sidx, c->samp_idx.as<int32_t>(), large_excl,
                                                                          samp_base, pair_off, c->ed.as<int32_t>(), in, p, stg,
                                                                          ncl_a, nmem_a, cnt + 10, phase);
        k_cluster<MID, SMALL><<<(unsigned)n_part, 64, cluster_lds(MID), s_mid>>>(n_part, c->part_start.as<int64_t>(), sidx, c->samp_idx.as<int32_t>(), large_excl,
                                                                             samp_base, pair_off, c->ed.as<int32_t>(), in, p, stg,
                                                                             ncl_a, nmem_a, cnt + 10, phase);
        k_cluster<SMALL, 0><<<(unsigned)n_part, 64, cluster_lds(SMALL), s_small>>>(n_part, c->part_start.as<int64_t>(), sidx, c->samp_idx.as<int32_t>(), large_excl,
                                                                               samp_base, pair_off, c->ed.as<int32_t>(), in, p, stg,
                                                                               ncl_a, nmem_a, cnt + 10, phase);
        HIPCHK(hipGetLastError());
        if (fork) {
            HIPCHK(hipEventRecord(c->ev[16], s_mid)); HIPCHK(hipEventRecord(c->ev[17], s_small));
            HIPCHK(hipStreamWaitEvent(ks, c->ev[16], 0)); HIPCHK(hipStreamWaitEvent(ks, c->ev[17], 0));
        }
        return SVX_OK;
    };
    SVXCHK(c->ed.reserve((size_t)(pair_total + 1) * 4));
    // partitions without insertions need nothing from the edit-distance rounds: their linkage runs beside them on a side stream
    const bool split = pair_total > 0;
    // ---- INS haplotype edit distances -----------------------------------------------------------------------------
    unsigned long long h_cnt[16] = {0};
    if (pair_total > 0) {
        if (!c->g_off_p) return svx_fail(SVX_E_STATE, "svx_set_genome must precede clustering of insertions", __FILE__, __LINE__, hipSuccess);
        SVXCHK(c->work.reserve((size_t)pair_total * sizeof(EditWork)));
        k_ins_pairs<<<(unsigned)n_part, 64, 0, st>>>(n_part, c->part_start.as<int64_t>(), sidx, c->samp_idx.as<int32_t>(), large_excl, pairs_a, pair_off, in, p,
                                                    c->work.as<EditWork>(), cnt + 8, pair_total);
        HIPCHK(hipGetLastError());
        // (the pair list is on the critical path, the side stream's linkage is not: its three launches are enqueued while k_ins_pairs runs)
        HIPCHK(hipEventRecord(c->ev[13], st));
        unsigned long long ticket = 0; const unsigned long long* words = nullptr;
        { MailSrc ms; memset(&ms, 0, sizeof ms); ms.k = 1; ms.p[0] = cnt; ms.n[0] = 16; SVXCHK(svx_mail_post(c, st, ms, &ticket)); }
        {
            HIPCHK(hipStreamWaitEvent(c->aux[4], c->ev[13], 0));
            SVXCHK(launch_cluster(c->aux[4], 1));
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev[14], c->aux[4]));
        }
        SVXCHK(svx_mail_wait(c, st, ticket, &words));
        memcpy(h_cnt, words, 16 * 8);
        const int64_t n_work = (int64_t)h_cnt[8];
        SVXCHK(c->cell_shards.reserve(1024 * 8));
        HIPCHK(hipMemsetAsync(c->cell_shards.p, 0, 1024 * 8, st));
        SVXCHK(svx_launch_edit_pairs(c, n_work, c->work.p, in, c->ed.as<int32_t>(), c->cell_shards.as<unsigned long long>()));
        S.n_edit_pairs = n_work;
    }
    HIPCHK(hipEventRecord(c->ev[10], st));
    // ---- per-partition clustering into the staging area ---------------------------------------------------------
    SVXCHK(launch_cluster(st, split ? 2 : 0));
    if (split) HIPCHK(hipStreamWaitEvent(st, c->ev[14], 0));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[11], st));
    // ---- dense output ------------------------------------------------------------------------------------------------
    SVXCHK(svx_exclusive_scan_i32_to_i64(c, ncl_a, clu_off, n_part + 1));
    SVXCHK(svx_exclusive_scan_i32_to_i64(c, nmem_a, mem_off, n_part + 1));
    int64_t tot2[2];
    SVXCHK(svx_mail_read3(c, st, clu_off + n_part, 1, &tot2[0], mem_off + n_part, 1, &tot2[1], cnt, 16, h_cnt));
    const int64_t ncl = tot2[0], nmem = tot2[1];
    const size_t CN = (size_t)(ncl + 1);
    // unsorted dense copy lives in tmp buffers, final tables in c->clu
    SVXCHK(c->tmp0.reserve(CN * (2 + 4 * 7 + 8 * 5 + 8) + 64));
    char* b = c->tmp0.as<char>();
    double* u_score = reinterpret_cast<double*>(b); b += CN * 8; double* u_span = reinterpret_cast<double*>(b); b += CN * 8;
    double* u_pos = reinterpret_cast<double*>(b); b += CN * 8; int64_t* u_moff = reinterpret_cast<int64_t*>(b); b += CN * 8;
    int64_t* u_part = reinterpret_cast<int64_t*>(b); b += CN * 8; uint64_t* u_key = reinterpret_cast<uint64_t*>(b); b += CN * 8;
    int32_t* u_contig = reinterpret_cast<int32_t*>(b); b += CN * 4; int32_t* u_start = reinterpret_cast<int32_t*>(b); b += CN * 4;
    int32_t* u_end = reinterpret_cast<int32_t*>(b); b += CN * 4; int32_t* u_contig2 = reinterpret_cast<int32_t*>(b); b += CN * 4;
    int32_t* u_start2 = reinterpret_cast<int32_t*>(b); b += CN * 4; int32_t* u_end2 = reinterpret_cast<int32_t*>(b); b += CN * 4;
    int32_t* u_size = reinterpret_cast<int32_t*>(b); b += CN * 4;
    uint8_t* u_type = reinterpret_cast<uint8_t*>(b); b += CN; uint8_t* u_aux = reinterpret_cast<uint8_t*>(b); b += CN;
    SVXCHK(c->tmp1.reserve((size_t)(nmem + 1) * 4));
    int32_t* u_members = c->tmp1.as<int32_t>();
    k_finalize<<<(unsigned)n_part, 64, 0, st>>>(n_part, samp_base, ncl_a, clu_off, mem_off, stg, u_type, u_aux, u_contig, u_start, u_end, u_contig2, u_start2,
                                               u_end2, u_score, u_span, u_pos, u_size, u_moff, u_part, u_key, rank);
    k_copy_members<<<(unsigned)n_part, 64, 0, st>>>(n_part, samp_base, nmem_a, mem_off, stg.members, u_members);
    HIPCHK(hipGetLastError());
    // final stable order
    SVXCHK(out.type.reserve(CN)); SVXCHK(out.aux.reserve(CN)); SVXCHK(out.contig.reserve(CN * 4)); SVXCHK(out.start.reserve(CN * 4));
    SVXCHK(out.end.reserve(CN * 4)); SVXCHK(out.contig2.reserve(CN * 4)); SVXCHK(out.start2.reserve(CN * 4)); SVXCHK(out.end2.reserve(CN * 4));
    SVXCHK(out.score.reserve(CN * 8)); SVXCHK(out.std_span.reserve(CN * 8)); SVXCHK(out.std_pos.reserve(CN * 8)); SVXCHK(out.size.reserve(CN * 4));
    SVXCHK(out.member_off.reserve((CN + 1) * 8)); SVXCHK(out.members.reserve((size_t)(nmem + 1) * 4)); SVXCHK(out.part_index.reserve(CN * 8));
    if (ncl > 0) {
        SVXCHK(c->tmp2.reserve(CN * 8 + CN * 4 * 2 + CN * 8));
        uint64_t* key2 = c->tmp2.as<uint64_t>(); uint32_t* perm0 = reinterpret_cast<uint32_t*>(key2 + CN); uint32_t* perm = perm0 + CN;
        int64_t* src_moff = reinterpret_cast<int64_t*>(perm + CN);
        k_iota_u32c<<<GRID(ncl, T), T, 0, st>>>(perm0, ncl);
        SVXCHK(svx_sort_pairs_u64(c, u_key, key2, perm0, perm, ncl, 0, 64));
        k_permute_clusters<<<GRID(ncl + 1, T), T, 0, st>>>(ncl, perm, u_type, u_aux, u_contig, u_start, u_end, u_contig2, u_start2, u_end2, u_score, u_span,
                                                          u_pos, u_size, u_moff, u_part, out.type.as<uint8_t>(), out.aux.as<uint8_t>(),
                                                          out.contig.as<int32_t>(), out.start.as<int32_t>(), out.end.as<int32_t>(),
                                                          out.contig2.as<int32_t>(), out.start2.as<int32_t>(), out.end2.as<int32_t>(),
                                                          out.score.as<double>(), out.std_span.as<double>(), out.std_pos.as<double>(),
                                                          out.size.as<int32_t>(), src_moff, out.part_index.as<int64_t>());
        SVXCHK(svx_exclusive_scan_i32_to_i64(c, out.size.as<int32_t>(), out.member_off.as<int64_t>(), ncl + 1));
        k_gather_members<<<GRID(ncl, 4), 256, 0, st>>>(ncl, out.size.as<int32_t>(), src_moff, out.member_off.as<int64_t>(), u_members, out.members.as<int32_t>());
        unsigned long long* tc = cnt + 0;
        HIPCHK(hipMemsetAsync(tc, 0, 6 * 8, st));
        k_type_counts<<<1, 64, 0, st>>>(ncl, out.type.as<uint8_t>(), tc);
        unsigned long long h_tc[6];
        SVXCHK(svx_mail_read(c, st, tc, 6, h_tc));
        for (int t = 0; t < SVX_NTYPES; t++) out.type_count[t] = (int64_t)h_tc[t];
    } else {
        HIPCHK(hipMemsetAsync(out.member_off.p, 0, 16, st));
    }
    out.n = ncl; out.n_members = nmem;
    HIPCHK(hipEventRecord(c->ev[12], st));
    HIPCHK(hipStreamSynchronize(st));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[8], c->ev[9])); S.t_partition_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[9], c->ev[10])); S.t_edit_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[10], c->ev[11])); S.t_linkage_ms = ms;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[8], c->ev[12])); S.t_cluster_ms = ms;
    S.n_partitions = n_part; S.n_large_partitions = n_large; S.n_pairs = (int64_t)h_cnt[10]; S.n_clusters = ncl;
    if (pair_total > 0 && c->cell_shards.p) {
        unsigned long long h_cells[1024];
        HIPCHK(hipMemcpy(h_cells, c->cell_shards.p, sizeof h_cells, hipMemcpyDeviceToHost));       // 8 KiB of stack: below the runtime's in-place pinning size (1 MiB; hostcopy.hpp), staged by the runtime
        unsigned long long tot = 0;
        for (int i = 0; i < 1024; i++) tot += h_cells[i];
        S.n_edit_cells = (int64_t)tot;
    }
    return SVX_OK;
}

int svx_cluster_impl(svx_ctx* c, const ClusterIn& in, int32_t n_contig, const int32_t* rank, const svx_params* pp) {
    c->xr_pending = c->xr_world > 1 && c->xr_fn;
    const int rc = svx_cluster_body(c, in, n_contig, rank, pp);
    if (rc != SVX_OK && c->xr_pending) {                    // the other ranks wait in a header exchange this rank will never reach
        const std::string keep = g_svx_err;
        svx_exchange_poison(c);
        g_svx_err = keep;
    }
    c->xr_pending = false;
    return rc;
}

// loads this translation unit's code object (HIP does it lazily, at the first launch): called by svx_ctx_create so that the first COLLECT / CLUSTER call
// of a context does not pay for it
void svx_preload_cluster() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_part_starts)); (void)hipGetLastError(); }
