// edit.hip - global (Needleman-Wunsch, unit cost) edit distance between insertion haplotypes on the GPU.
//
// Replaces compute_haplotype_edit_distance (src/svim/SVIM_clustering.py:32-45): the reference builds
//   hap_k = ref[ws:start_k] + inserted_k + ref[start_k:we]   (window = min/max start -+ 100, clipped to the contig)
// as Python strings with four FASTA fetches per pair and calls edlib.align(h1, h2)["editDistance"].  Here the
// haplotypes are never materialised: a `Hap` is a virtual concatenation of three byte ranges living in HBM
// (genome codes + the signature's inserted bases), the common prefix/suffix is stripped wave-parallel, and the
// remaining core is solved exactly with Myers/Hyyro bit-vector DP laid out as a 64-lane systolic array:
//   lane l owns R consecutive 32-row blocks of the shorter string (bit-sliced into 4 planes, 4-bit alphabet),
//   at step t it processes text column t-l, receives (symbol, horizontal delta) from lane l-1 through a single
//   DPP wave shift and hands its own to lane l+1.  Text symbols are fetched 64 at a time (one coalesced load,
//   prefetched a chunk ahead) and broadcast with v_readlane.  No LDS, no MFMA: integer ALU bound.
#include "common.hpp"

struct Hap {
    const uint8_t* p0; const uint8_t* p1; const uint8_t* p2;
    int n0, n1, n2, len;
    __device__ __forceinline__ uint32_t at(int i) const {
        if (i < n0) return p0[i];
        i -= n0;
        if (i < n1) return p1[i];
        return p2[i - n1];
    }
};

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

// one 32-row block, one text column
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

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t v) {
    // lane l receives lane l-1's value, lane 0 receives 0 (gfx9 DPP control wave_shr:1)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}

// Exact distance between the cores pat (length m >= 1, the shorter) and txt (length n >= m); whole wave cooperates.
// R = 32-row blocks per lane held in registers (m <= 64*32*R).
template <int R>
__device__ int systolic_distance(const Hap& pat, int poff, int m, const Hap& txt, int toff, int n) {
    const int lane = lane_id();
    const int nb = (m + 31) >> 5;
    uint32_t pl[R][4], vm[R], pv[R], mv[R], top[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int blk = lane * R + r;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, v = 0;
        if (blk < nb) {
            const int row0 = blk << 5;
            for (int i = 0; i < 32; i++) {
                const int row = row0 + i;
                if (row < m) {
                    const uint32_t c = pat.at(poff + row);
                    a0 |= (c & 1u) << i; a1 |= ((c >> 1) & 1u) << i; a2 |= ((c >> 2) & 1u) << i; a3 |= ((c >> 3) & 1u) << i;
                    v |= 1u << i;
                }
            }
        }
        pl[r][0] = a0; pl[r][1] = a1; pl[r][2] = a2; pl[r][3] = a3; vm[r] = v;
        pv[r] = 0xffffffffu; mv[r] = 0u;
        top[r] = (blk == nb - 1) ? (1u << ((m - 1) & 31)) : 0x80000000u;
    }
    const int lanes_used = (nb + R - 1) / R;
    const int last_lane = lanes_used - 1;
    const int last_r = (nb - 1) - last_lane * R;
    int score = m;
    const int steps = n + lanes_used - 1;
    uint32_t out = 0;
    uint32_t tc_next = (lane < n) ? txt.at(toff + lane) : 0u;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const uint32_t tc = tc_next;
        const int nxt = t0 + 64 + lane;
        tc_next = (nxt < n) ? txt.at(toff + nxt) : 0u;
#pragma unroll 4
        for (int j = 0; j < 64; j++) {
            const int t = t0 + j;
            if (t >= steps) break;
            uint32_t in = dpp_wave_shr1(out);
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)tc, j);
            if (lane == 0) in = (t < n) ? (c0 | 0x10u | 0x40u) : 0u;       // top row: horizontal delta +1
            if ((in & 0x40u) && lane < lanes_used) {
                const uint32_t c = in & 15u;
                const uint32_t n0 = (c & 1u) - 1u, n1 = ((c >> 1) & 1u) - 1u, n2 = ((c >> 2) & 1u) - 1u, n3 = ((c >> 3) & 1u) - 1u;
                uint32_t hp = (in >> 4) & 1u, hm = (in >> 5) & 1u;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (lane * R + r < nb) {
                        const uint32_t eq = (pl[r][0] ^ n0) & (pl[r][1] ^ n1) & (pl[r][2] ^ n2) & (pl[r][3] ^ n3) & vm[r];
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

// Fallback for cores longer than 64*32*8 rows: same systolic schedule, block state in a global scratch area.
__device__ int systolic_distance_big(const Hap& pat, int poff, int m, const Hap& txt, int toff, int n, uint32_t* scratch) {
    const int lane = lane_id();
    const int nb = (m + 31) >> 5;
    const int R = (nb + 63) / 64;
    // scratch layout per block: pl0..pl3, vm, pv, mv  (7 words), block-major
    for (int r = 0; r < R; r++) {
        const int blk = lane * R + r;
        if (blk >= nb) break;
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, v = 0;
        const int row0 = blk << 5;
        for (int i = 0; i < 32; i++) {
            const int row = row0 + i;
            if (row < m) {
                const uint32_t c = pat.at(poff + row);
                a0 |= (c & 1u) << i; a1 |= ((c >> 1) & 1u) << i; a2 |= ((c >> 2) & 1u) << i; a3 |= ((c >> 3) & 1u) << i;
                v |= 1u << i;
            }
        }
        uint32_t* s = scratch + (size_t)blk * 7;
        s[0] = a0; s[1] = a1; s[2] = a2; s[3] = a3; s[4] = v; s[5] = 0xffffffffu; s[6] = 0u;
    }
    const int lanes_used = (nb + R - 1) / R;
    const int last_lane = lanes_used - 1;
    int score = m;
    const int steps = n + lanes_used - 1;
    uint32_t out = 0;
    uint32_t tc_next = (lane < n) ? txt.at(toff + lane) : 0u;
    for (int t0 = 0; t0 < steps; t0 += 64) {
        const uint32_t tc = tc_next;
        const int nxt = t0 + 64 + lane;
        tc_next = (nxt < n) ? txt.at(toff + nxt) : 0u;
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
                    uint32_t* s = scratch + (size_t)blk * 7;
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

#define SVX_EDIT_BIG (-2)      // returned by edit_core when the pair needs the scratch-backed path

// Whole-wave edit distance of two virtual strings.  cells: (optional) DP-cell count for statistics.
__device__ int edit_core(const Hap& A, const Hap& B, uint32_t* big_scratch, unsigned long long* cells, int* big_m = nullptr) {
    const int lane = lane_id();
    const int la = A.len, lb = B.len;
    const int mn = la < lb ? la : lb;
    // common prefix
    int pre = 0;
    while (pre < mn) {
        const int i = pre + lane;
        const bool diff = (i >= mn) || (A.at(i) != B.at(i));
        const unsigned long long d = __ballot(diff);
        if (d) { pre += __ffsll((long long)d) - 1; break; }
        pre += 64;
    }
    if (pre > mn) pre = mn;
    // common suffix (not overlapping the prefix)
    int suf = 0;
    const int lim = mn - pre;
    while (suf < lim) {
        const int i = suf + lane;
        const bool diff = (i >= lim) || (A.at(la - 1 - i) != B.at(lb - 1 - i));
        const unsigned long long d = __ballot(diff);
        if (d) { suf += __ffsll((long long)d) - 1; break; }
        suf += 64;
    }
    if (suf > lim) suf = lim;
    const int ca = la - pre - suf, cb = lb - pre - suf;
    if (ca == 0 || cb == 0) return ca + cb;
    const bool a_short = ca <= cb;
    const Hap& pat = a_short ? A : B;
    const Hap& txt = a_short ? B : A;
    const int m = a_short ? ca : cb, n = a_short ? cb : ca;
    if (cells && lane == 0) atomicAdd(cells, (unsigned long long)m * (unsigned long long)n);
    const int nb = (m + 31) >> 5;
    if (nb <= 64) return systolic_distance<1>(pat, pre, m, txt, pre, n);
    if (nb <= 128) return systolic_distance<2>(pat, pre, m, txt, pre, n);
    if (nb <= 256) return systolic_distance<4>(pat, pre, m, txt, pre, n);
    if (nb <= 512) return systolic_distance<8>(pat, pre, m, txt, pre, n);
    if (!big_scratch) { if (big_m) *big_m = m; return SVX_EDIT_BIG; }
    return systolic_distance_big(pat, pre, m, txt, pre, n, big_scratch);
}

// ---- kernels -------------------------------------------------------------------------------------------

// plain string pairs (svx_edit_distance test/utility entry point)
__global__ __launch_bounds__(256) void k_edit_plain(long long n_pairs, const uint8_t* codes, const int64_t* a_off, const int64_t* b_off,
                                                    int32_t* out, uint32_t* scratch, long long scratch_words_per_wave) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_pairs) return;
    const Hap A = plain_hap(codes + a_off[w], (int)(a_off[w + 1] - a_off[w]));
    const Hap B = plain_hap(codes + b_off[w], (int)(b_off[w + 1] - b_off[w]));
    const int d = edit_core(A, B, scratch ? scratch + (size_t)w * scratch_words_per_wave : nullptr, nullptr);
    if (lane_id() == 0) out[w] = d;
}

// insertion-signature pairs: work item = (global signature index a, b, output slot)
struct EditWork { uint32_t a, b; long long slot; };

__global__ __launch_bounds__(256) void k_edit_pairs(long long n_work, const EditWork* work, ClusterIn in, const int64_t* g_off,
                                                    const uint8_t* g_codes, int32_t* ed, unsigned long long* cells,
                                                    unsigned long long* big_count, uint2* big_list) {
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_work) return;
    const EditWork wk = work[w];
    const long long s1 = in.start[wk.a], s2 = in.start[wk.b];
    const long long ws = (s1 < s2 ? s1 : s2) - 100, we = (s1 > s2 ? s1 : s2) + 100;
    const Hap A = make_hap(g_off, g_codes, in.contig[wk.a], s1, in.seq + in.seq_off[wk.a], (int)(in.seq_off[wk.a + 1] - in.seq_off[wk.a]), ws, we);
    const Hap B = make_hap(g_off, g_codes, in.contig[wk.b], s2, in.seq + in.seq_off[wk.b], (int)(in.seq_off[wk.b + 1] - in.seq_off[wk.b]), ws, we);
    int big_m = 0;
    const int d = edit_core(A, B, nullptr, cells, &big_m);
    if (lane_id() == 0) {
        if (d == SVX_EDIT_BIG) { const unsigned long long i = atomicAdd(big_count, 1ull); big_list[i] = make_uint2((uint32_t)w, (uint32_t)big_m); }
        else ed[wk.slot] = d;
    }
}

// second pass for the (rare) pairs whose shorter core exceeds 16384 symbols
__global__ __launch_bounds__(64) void k_edit_pairs_big(long long n_big, const uint2* big_list, const long long* scratch_off, const EditWork* work,
                                                       ClusterIn in, const int64_t* g_off, const uint8_t* g_codes, int32_t* ed, uint32_t* scratch) {
    const long long q = blockIdx.x;
    if (q >= n_big) return;
    const EditWork wk = work[big_list[q].x];
    const long long s1 = in.start[wk.a], s2 = in.start[wk.b];
    const long long ws = (s1 < s2 ? s1 : s2) - 100, we = (s1 > s2 ? s1 : s2) + 100;
    const Hap A = make_hap(g_off, g_codes, in.contig[wk.a], s1, in.seq + in.seq_off[wk.a], (int)(in.seq_off[wk.a + 1] - in.seq_off[wk.a]), ws, we);
    const Hap B = make_hap(g_off, g_codes, in.contig[wk.b], s2, in.seq + in.seq_off[wk.b], (int)(in.seq_off[wk.b + 1] - in.seq_off[wk.b]), ws, we);
    const int d = edit_core(A, B, scratch + scratch_off[q], nullptr);
    if (lane_id() == 0) ed[wk.slot] = d;
}

int svx_edit_distance_pairs(svx_ctx* c, int64_t n_pairs, const uint8_t* codes_dev, const int64_t* a_off_dev, const int64_t* b_off_dev,
                            int32_t* out_dev) {
    if (n_pairs <= 0) return SVX_OK;
    // utility path: always provide scratch large enough for the longest string (sizes are read back once)
    std::vector<int64_t> ha((size_t)n_pairs + 1), hb((size_t)n_pairs + 1);
    HIPCHK(hipMemcpyAsync(ha.data(), a_off_dev, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(hb.data(), b_off_dev, (size_t)(n_pairs + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t mx = 0;
    for (int64_t i = 0; i < n_pairs; i++) {
        const int64_t la = ha[i + 1] - ha[i], lb = hb[i + 1] - hb[i];
        const int64_t m = la < lb ? la : lb;
        if (m > mx) mx = m;
    }
    uint32_t* scratch = nullptr;
    long long words = 0;
    if (mx > 64 * 32 * 8) {
        words = ((mx + 31) / 32) * 7;
        SVXCHK(c->tmp5.reserve((size_t)n_pairs * (size_t)words * 4));
        scratch = c->tmp5.as<uint32_t>();
    }
    k_edit_plain<<<(unsigned)((n_pairs + 3) / 4), 256, 0, c->stream>>>(n_pairs, codes_dev, a_off_dev, b_off_dev, out_dev, scratch, words);
    HIPCHK(hipGetLastError());
    return SVX_OK;
}

// used by cluster.hip
int svx_launch_edit_pairs(svx_ctx* c, int64_t n_work, const void* work_dev, const ClusterIn& in, int32_t* ed_dev,
                          unsigned long long* cells_dev) {
    if (n_work <= 0) return SVX_OK;
    hipStream_t st = c->stream;
    SVXCHK(c->tmp4.reserve(16 + (size_t)n_work * 8));
    unsigned long long* big_count = c->tmp4.as<unsigned long long>();
    uint2* big_list = reinterpret_cast<uint2*>(big_count + 2);
    HIPCHK(hipMemsetAsync(big_count, 0, 16, st));
    k_edit_pairs<<<(unsigned)((n_work + 3) / 4), 256, 0, st>>>(n_work, (const EditWork*)work_dev, in, c->g_off_p, c->g_codes_p, ed_dev, cells_dev,
                                                              big_count, big_list);
    HIPCHK(hipGetLastError());
    unsigned long long nbig = 0;
    HIPCHK(hipMemcpyAsync(&nbig, big_count, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (nbig) {
        // rare: shorter core > 16384 symbols.  Size the block-state scratch exactly from the recorded core lengths.
        std::vector<uint2> items((size_t)nbig);
        HIPCHK(hipMemcpyAsync(items.data(), big_list, (size_t)nbig * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<long long> off((size_t)nbig + 1, 0);
        for (size_t i = 0; i < (size_t)nbig; i++) off[i + 1] = off[i] + (((long long)items[i].y + 31) / 32) * 7;
        SVXCHK(c->tmp5.reserve((size_t)off[nbig] * 4 + 16));
        SVXCHK(c->tmp3.reserve((size_t)(nbig + 1) * 8));
        HIPCHK(hipMemcpyAsync(c->tmp3.p, off.data(), (size_t)(nbig + 1) * 8, hipMemcpyHostToDevice, st));
        k_edit_pairs_big<<<(unsigned)nbig, 64, 0, st>>>((long long)nbig, big_list, c->tmp3.as<long long>(), (const EditWork*)work_dev, in, c->g_off_p,
                                                       c->g_codes_p, ed_dev, c->tmp5.as<uint32_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
    }
    return SVX_OK;
}
