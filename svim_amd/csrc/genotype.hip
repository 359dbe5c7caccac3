// genotype.hip - reads supporting the reference allele around every candidate.
//
// Replaces the inner loop of genotype() (src/svim/SVIM_genotyping.py:34-93): for every candidate the reference re-opens the
// BAM index (bam.fetch(contig, start-1000, end+1000), :48), walks at most 500 eligible alignments (:56-68: not a read of the
// variant, mapped, not secondary, mapq >= min_mapq) and collects the names of those that span the locus (:70-77).  Here the
// alignment records are a structure-of-arrays in HBM in file order; one wavefront per candidate binary-searches its window and
// walks it 64 records at a time.  HBM-bound (20 B per record visited), latency-dominated for the short walks of real data.
#include "common.hpp"
#include "hostcopy.hpp"

struct AlnIndexDev {
    int64_t n; int32_t n_contig;
    const int64_t* contig_first; const int64_t* contig_len;
    const int32_t* pos; const int32_t* end; const int32_t* end_prefmax; const uint16_t* flag; const uint8_t* mapq; const int32_t* name_id;
};

// running maximum of end_or_pos1 inside every contig: the first record that can overlap a window start is found by bisection
__global__ __launch_bounds__(64) void k_end_prefmax(AlnIndexDev ix, int32_t* prefmax) {       // one wave per contig, 64 records per step
    const int c = blockIdx.x;
    if (c >= ix.n_contig) return;
    const int lane = lane_id();
    const long long first = ix.contig_first[c], last = ix.contig_first[c + 1];
    int32_t carry = INT32_MIN;
    for (long long base = first; base < last; base += 64) {
        const long long i = base + lane;
        int32_t v = INT32_MIN;
        if (i < last) { v = ix.end[i]; if (v <= ix.pos[i]) v = ix.pos[i] + 1; }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int32_t u = __shfl_up(v, o, 64); if (lane >= o && u > v) v = u; }
        if (carry > v) v = carry;
        if (i < last) prefmax[i] = v;
        carry = __shfl(v, 63, 64);
    }
}

#define GENO_LIMIT 500

__global__ __launch_bounds__(64) void k_genotype(AlnIndexDev ix, int mode, long long n_cand, const int32_t* cand_tid, const int32_t* cand_start,
                                                 const int32_t* cand_end, const int64_t* member_off, const int32_t* member_names, int min_mapq,
                                                 int32_t* out_ref) {
    __shared__ int32_t names[GENO_LIMIT + 64];
    const long long c = blockIdx.x;
    if (c >= n_cand) return;
    const int lane = lane_id();
    const int tid = cand_tid[c];
    if (tid < 0 || tid >= ix.n_contig) { if (lane == 0) out_ref[c] = 0; return; }
    const long long start = cand_start[c], end = cand_end[c];
    const long long clen = ix.contig_len[tid];
    const long long ws = start - 1000 > 0 ? start - 1000 : 0, we = end + 1000 < clen ? end + 1000 : clen;      // :48
    const long long first = ix.contig_first[tid], last = ix.contig_first[tid + 1];
    // records [i_lo, i_hi): the first whose running end maximum exceeds ws ... the first with pos >= we
    long long lo = first, hi = last;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)ix.end_prefmax[mid] > ws) hi = mid; else lo = mid + 1; }
    const long long i_lo = lo;
    lo = first; hi = last;
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)ix.pos[mid] >= we) hi = mid; else lo = mid + 1; }
    const long long i_hi = ws < we ? lo : i_lo;
    const long long m0 = member_off[c], mn = member_off[c + 1] - m0;
    const long long mo2 = end - start < 4000 ? end - start : 4000;         // 2 * min((end - start) / 2, 2000)  (:71)
    int counted = 0, n_names = 0;
    for (long long base = i_lo; base < i_hi && counted < GENO_LIMIT; base += 64) {
        const long long i = base + lane;
        bool pass = false, support = false;
        int32_t name = -1;
        if (i < i_hi) {
            const long long rs = ix.pos[i];
            long long re = ix.end[i];
            const long long endp = re > rs ? re : rs + 1;
            if (endp > ws) {                                               // overlaps the fetched region
                name = ix.name_id[i];
                long long a = 0, b = mn;                                   // current_alignment.query_name in reads_supporting_variant (:63)
                while (a < b) { const long long mid = (a + b) >> 1; if (member_names[m0 + mid] < name) a = mid + 1; else b = mid; }
                const bool in_variant = a < mn && member_names[m0 + a] == name;
                const unsigned flag = ix.flag[i];
                pass = !in_variant && !(flag & 0x4u) && !(flag & 0x100u) && (int)ix.mapq[i] >= min_mapq;      // :65
                if (mode == 0) support = (2 * rs < 2 * end - mo2 && re > end + 100) || (rs < start - 100 && 2 * re > 2 * start + mo2);   // :72-74
                else support = rs < start - 100 && re > end + 100;                                                                      // :76
            }
        }
        const unsigned long long pm = __ballot(pass);
        const int ordinal = counted + (int)__popcll(pm & lanemask_lt()) + 1;          // aln_no after this alignment (:68)
        const bool take = pass && ordinal <= GENO_LIMIT && support;
        const unsigned long long tm = __ballot(take);
        if (take) names[n_names + (int)__popcll(tm & lanemask_lt())] = name;
        n_names += (int)__popcll(tm);
        counted += (int)__popcll(pm);
    }
    __syncthreads();
    // len(set(names))
    int distinct = 0;
    for (int j = lane; j < n_names; j += 64) {
        const int32_t v = names[j];
        bool first_seen = true;
        for (int k = 0; k < j; k++) if (names[k] == v) { first_seen = false; break; }
        distinct += first_seen;
    }
    distinct = wave_sum_i32(distinct);
    if (lane == 0) out_ref[c] = distinct;
}

int svx_set_alignment_index_impl(svx_ctx* c, const svx_aln_index* h) {
    hipStream_t st = c->stream;
    const size_t n = (size_t)h->n, nc = (size_t)h->n_contig;
    DevBuf* b = c->geno;
    struct Up { DevBuf* d; const void* src; size_t bytes; } ups[] = {
        {&b[0], h->contig_first, (nc + 1) * 8}, {&b[1], h->contig_len, nc * 8}, {&b[2], h->pos, n * 4}, {&b[3], h->end, n * 4},
        {&b[4], h->flag, n * 2}, {&b[5], h->mapq, n}, {&b[6], h->name_id, n * 4}};
    for (auto& u : ups) {
        SVXCHK(u.d->reserve(u.bytes + 64));
        SVXCHK(svx_h2d(u.d->p, u.src, u.bytes, st));
    }
    SVXCHK(b[7].reserve(n * 4 + 64));
    c->geno_n = h->n; c->geno_contigs = h->n_contig;
    AlnIndexDev ix{h->n, h->n_contig, b[0].as<int64_t>(), b[1].as<int64_t>(), b[2].as<int32_t>(), b[3].as<int32_t>(), b[7].as<int32_t>(),
                   b[4].as<uint16_t>(), b[5].as<uint8_t>(), b[6].as<int32_t>()};
    if (h->n_contig > 0) k_end_prefmax<<<(unsigned)h->n_contig, 64, 0, st>>>(ix, b[7].as<int32_t>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return SVX_OK;
}

int svx_genotype_impl(svx_ctx* c, int32_t mode, int64_t n_cand, const int32_t* tid, const int32_t* start, const int32_t* end, const int64_t* moff,
                      const int32_t* mnames, int32_t min_mapq, int32_t* out) {
    if (c->geno_contigs < 0) return svx_fail(SVX_E_STATE, "svx_set_alignment_index must precede svx_genotype", __FILE__, __LINE__, hipSuccess);
    if (n_cand <= 0) return SVX_OK;
    hipStream_t st = c->stream;
    const size_t n = (size_t)n_cand, nm = (size_t)moff[n_cand];
    DevBuf* b = c->geno;
    SVXCHK(b[8].reserve(n * 4 * 4 + 64)); SVXCHK(b[9].reserve((n + 1) * 8 + 64)); SVXCHK(b[10].reserve(nm * 4 + 64));
    int32_t* d_tid = b[8].as<int32_t>(); int32_t* d_start = d_tid + n; int32_t* d_end = d_start + n; int32_t* d_out = d_end + n;
    HostCopy hc(st);
    SVXCHK(hc.h2d(d_tid, tid, n * 4)); SVXCHK(hc.h2d(d_start, start, n * 4)); SVXCHK(hc.h2d(d_end, end, n * 4));
    SVXCHK(hc.h2d(b[9].p, moff, (n + 1) * 8)); SVXCHK(hc.h2d(b[10].p, mnames, nm * 4));
    AlnIndexDev ix{c->geno_n, c->geno_contigs, b[0].as<int64_t>(), b[1].as<int64_t>(), b[2].as<int32_t>(), b[3].as<int32_t>(), b[7].as<int32_t>(),
                   b[4].as<uint16_t>(), b[5].as<uint8_t>(), b[6].as<int32_t>()};
    k_genotype<<<(unsigned)n_cand, 64, 0, st>>>(ix, mode, n_cand, d_tid, d_start, d_end, b[9].as<int64_t>(), b[10].as<int32_t>(), min_mapq, d_out);
    HIPCHK(hipGetLastError());
    SVXCHK(hc.d2h(out, d_out, n * 4));
    return hc.finish();
}
