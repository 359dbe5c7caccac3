// common.hpp - context, device buffers and wave helpers shared by the libsvx translation units (gfx950 only).
#pragma once
#include <chrono>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/svx.h"

#define SVX_FLAG_USED_MASK (SVX_FLAG_SKIP | 4u | 256u)

extern thread_local std::string g_svx_err;
int svx_fail(int code, const char* what, const char* file, int line, hipError_t e);

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return svx_fail(SVX_E_HIP, #expr, __FILE__, __LINE__, e_);               \
    } while (0)
#define SVXCHK(expr)                                                                                   \
    do {                                                                                               \
        int rc_ = (expr);                                                                              \
        if (rc_ != SVX_OK) return rc_;                                                                 \
    } while (0)

// Growable device allocation (never shrinks; contents are NOT preserved on growth unless asked).
// time spent in hipMalloc / number of calls / bytes (SVX_ALLOC_STATS=1 prints them when a context is destroyed)
inline double& svx_alloc_seconds() { static double v = 0; return v; }
inline long long& svx_alloc_calls() { static long long v = 0; return v; }
inline size_t& svx_alloc_bytes() { static size_t v = 0; return v; }
// Guard mode (SVX_ALLOC_GUARD=1, debugging): every buffer is a mapping of its own whose LAST byte asked for (rounded up to 16) is the last mapped byte, with
// unmapped address space on both sides - an access more than 15 bytes behind what reserve() was asked for is a GPU fault at once instead of a read of whatever
// the allocator placed there (api.hip: svx_guard_alloc).  The 12.5 % growth slack is not added in this mode.
bool svx_guard_mode();
bool svx_guard_slack();          // SVX_ALLOC_GUARD_SLACK=1: the guard sits behind the growth slack (same reallocation pattern as without guard; only far overruns fault)
int svx_guard_alloc(void** out, size_t bytes);
void svx_guard_free(void* p);
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes, bool keep = false, hipStream_t s = nullptr) {
        if (bytes <= cap) return SVX_OK;
        const bool guard = svx_guard_mode();
        size_t ncap = guard && !svx_guard_slack() ? bytes : bytes + bytes / 8 + 256;
        void* np = nullptr;
        const auto t0_ = std::chrono::steady_clock::now();
        if (guard) SVXCHK(svx_guard_alloc(&np, ncap)); else HIPCHK(hipMalloc(&np, ncap));
        svx_alloc_seconds() += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); svx_alloc_calls()++; svx_alloc_bytes() += ncap;
        if (keep && p && cap) {
            HIPCHK(hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
        release();
        p = np;
        cap = ncap;
        return SVX_OK;
    }
    void release() {
        if (p) { if (svx_guard_mode()) svx_guard_free(p); else (void)hipFree(p); }
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Device-resident signature table (SoA); `cap` entries allocated.
struct DevSigs {
    int64_t n = 0, cap = 0;
    DevBuf key, type, src, aux, contig, start, end, contig2, pos2, read_id, rec, qpos, qlen;
    DevBuf seq_off, seq;          // filled by the gather step (per-signature inserted bases)
    int64_t n_seq = 0;
    int reserve(int64_t c) {
        if (c <= cap) return SVX_OK;
        SVXCHK(key.reserve(c * 8)); SVXCHK(type.reserve(c)); SVXCHK(src.reserve(c)); SVXCHK(aux.reserve(c));
        SVXCHK(contig.reserve(c * 4)); SVXCHK(start.reserve(c * 4)); SVXCHK(end.reserve(c * 4));
        SVXCHK(contig2.reserve(c * 4)); SVXCHK(pos2.reserve(c * 4)); SVXCHK(read_id.reserve(c * 4));
        SVXCHK(rec.reserve(c * 4)); SVXCHK(qpos.reserve(c * 4)); SVXCHK(qlen.reserve(c * 4));
        cap = c;
        return SVX_OK;
    }
    // grow to at least c entries KEEPING the first n (accumulator of several batches); rec / qpos / qlen are per-batch scratch and not kept
    int reserve_keep(int64_t c, hipStream_t s) {
        if (c <= cap) return SVX_OK;
        const int64_t nc = c + c / 2 + 1024;
        DevBuf* b8[] = {&key}; DevBuf* b1[] = {&type, &src, &aux}; DevBuf* b4[] = {&contig, &start, &end, &contig2, &pos2, &read_id};
        for (auto* b : b8) SVXCHK(b->reserve((size_t)nc * 8, true, s));
        for (auto* b : b1) SVXCHK(b->reserve((size_t)nc, true, s));
        for (auto* b : b4) SVXCHK(b->reserve((size_t)nc * 4, true, s));
        SVXCHK(seq_off.reserve((size_t)(nc + 2) * 8, true, s));
        cap = nc;
        return SVX_OK;
    }
    void release() {
        DevBuf* all[] = {&key, &type, &src, &aux, &contig, &start, &end, &contig2, &pos2, &read_id, &rec, &qpos, &qlen, &seq_off, &seq};
        for (auto* b : all) b->release();
        n = cap = 0;
    }
};

// Raw pointer view of a signature table used by kernels.
struct SigPtrs {
    uint64_t* key; uint8_t* type; uint8_t* src; uint8_t* aux;
    int32_t *contig, *start, *end, *contig2, *pos2, *read_id, *rec, *qpos, *qlen;
};
inline SigPtrs sig_ptrs(const DevSigs& d) {
    SigPtrs p;
    p.key = d.key.as<uint64_t>(); p.type = d.type.as<uint8_t>(); p.src = d.src.as<uint8_t>(); p.aux = d.aux.as<uint8_t>();
    p.contig = d.contig.as<int32_t>(); p.start = d.start.as<int32_t>(); p.end = d.end.as<int32_t>();
    p.contig2 = d.contig2.as<int32_t>(); p.pos2 = d.pos2.as<int32_t>(); p.read_id = d.read_id.as<int32_t>();
    p.rec = d.rec.as<int32_t>(); p.qpos = d.qpos.as<int32_t>(); p.qlen = d.qlen.as<int32_t>();
    return p;
}

// Read-only view of the clustering input (device pointers).
struct ClusterIn {
    int64_t n;
    const uint8_t* type; const uint8_t* aux;
    const int32_t *contig, *start, *end, *contig2, *pos2, *read_id;
    const int64_t* seq_off; const uint8_t* seq;
};

struct DevClusters {
    int64_t n = 0, n_members = 0, cap = 0;
    int64_t type_count[SVX_NTYPES] = {0, 0, 0, 0, 0, 0};
    DevBuf type, contig, start, end, contig2, start2, end2, aux, score, std_span, std_pos, size, member_off, members, part_index;
};

#define SVX_N_AUX 8

struct svx_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev[24];
    hipStream_t aux[SVX_N_AUX];     // side streams of the edit-distance pipeline: [0..1] band classes, [2..3] full-matrix classes (the high-priority ones: api.hip); [4] low: linkage of the partitions that need no edit distances; [5] high: the widest band classes of round 0; [6] normal priority (like the main stream): the packed haplotype store built ahead of the pair list (A/B); [7] high: the row-block launches (SVX_EDIT_BLOCKED)
    // device copies of a host-resident batch
    std::vector<DevBuf> batch_bufs;
    // COLLECT results
    DevSigs sig, bnd;               // sorted by key
    // svx_collect_accumulate: results of successive svx_collect calls (the batches of one input file) appended on the device
    DevSigs acc_sig, acc_bnd; bool accumulate = false; uint64_t slot_base = 0;
    DevSigs raw_sig, raw_bnd;       // unordered emission buffers
    DevBuf counters;                // device counters (uint64 x 16)
    DevBuf shard_cnt;               // per-shard allocation counters + prefix of the raw indel buffer
    DevBuf raw_indel;               // RawIndel records written by the scan kernel
    DevBuf rec_geom, seg_geom;      // int32 x 5 per record / per segment row
    DevBuf seg_ws;                  // segment analysis workspace
    DevBuf tmp0, tmp1, tmp2, tmp3, tmp4, tmp5, sort_tmp, scan_tmp;
    // genome
    DevBuf g_off, g_codes; int32_t g_n = 0; bool g_borrowed = false; const int64_t* g_off_p = nullptr; const uint8_t* g_codes_p = nullptr;
    // CLUSTER workspace + results
    DevBuf user_sig[12]; DevBuf c_rank;
    DevBuf k_hi, k_lo, k_idx, k_hi2, k_lo2, k_idx2, part_flag, part_id, part_start, part_meta, samp_idx, large_list, samp_stream;
    DevBuf cell_shards;
    DevBuf geno[11]; int64_t geno_n = 0; int32_t geno_contigs = -1;      // GENOTYPE: resident alignment index + per-call candidate buffers
    DevBuf samp_meta, samp_table, samp_runs, samp_chain;    // consumption tables of the sampling walk; per-type stream positions
    int xr_rank = 0, xr_world = 1; svx_allgather_fn xr_fn = nullptr; void* xr_user = nullptr; bool xr_pending = false;   // svx_cluster_set_ranks
    long long stream_start[SVX_NTYPES] = {0, 0, 0, 0, 0, 0}, stream_end[SVX_NTYPES] = {0, 0, 0, 0, 0, 0};              // of the last svx_cluster
    DevBuf mt_words; long long mt_have = 0;      // tempered MT19937 words after seed(1524), kept across calls
    DevBuf pair_off, ed, work, stage, stage_members, labels;
    DevBuf e_words, e_off, e_scratch, e_rec, e_desc, e_key, e_val, e_slot, e_fail, e_big_list, e_big_state, e_big_off;
    DevBuf e_blk[2];                // boundary words + hand-over values of the row-block route (SVX_EDIT_BLOCKED), per alphabet
    DevBuf e_retry[3];              // per-class retry lists of the edit-distance rounds (rotating)     // edit-distance pipeline
    DevClusters clu;
    // Band speculation: a pair without a useful distance bound starts in the band sized for edit_guess * (core length) differences beyond
    // the length gap.  Chosen per call from a sample of that call's own pairs (k_edit_pilot, edit.hip) - no state survives a call;
    // 0.125 is only the fallback for calls too small to sample.  Routing only: results never depend on it.  SVX_EDIT_GUESS=<fraction> pins it.
    float edit_guess = 0.125f; bool edit_guess_pinned = false; float edit_guess_last = 0.125f;
    // Packed haplotype store built AHEAD of the pair list, on a side stream beside the partition / sampling kernels (svx_edit_prepack_*):
    // 0 = none, 1 = word counts + offsets enqueued, 2 = packed (ev[18] marks the end)
    int prepack_state = 0; long long prepack_radius = 0, prepack_n = 0; DevBuf prepack_tmp;
    int64_t* pinned = nullptr;       // 4 KB of pinned host memory for small asynchronous read-backs
    // Mailbox (api.hip, svx_mail_post / svx_mail_wait): the few words a host decision waits for in the middle of a call (list sizes, class bounds, retry
    // counters) are written INTO page-locked host memory by a one-wave kernel at the end of the producing stream, behind a sequence word the host spins on -
    // no copy command, no interrupt-driven wake-up.  SVX_MAILBOX=0: hipMemcpyAsync + hipStreamSynchronize (A/B switch).
    unsigned long long* mail = nullptr; unsigned long long mail_seq = 0; int mail_mode = 1;
    DevBuf e_hist;
    bool edit_force_full = false;  // debugging aid (env SVX_EDIT_FORCE_FULL=1): every pair through the full-matrix kernel
    bool no_seq_gather = false;   // svx_cigar_indel hook: positions only
    svx_stats stats;
    int64_t last_cluster_source_n = 0;
};

// ---- primitives (prims.hip, scan.hpp: hand-written radix sort and scan) -------------------------------
int svx_sort_pairs_u64(svx_ctx* c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                       int64_t n, int begin_bit, int end_bit);
int svx_exclusive_scan_i64(svx_ctx* c, const int64_t* in, int64_t* out, int64_t n);   // out[n] NOT written
int svx_exclusive_scan_i64_on(svx_ctx* c, const int64_t* in, int64_t* out, int64_t n, hipStream_t stream, DevBuf& tmp);
int svx_exclusive_scan_i32_to_i64(svx_ctx* c, const int32_t* in, int64_t* out, int64_t n);

#define SVX_MAIL_SLOTS 16
#define SVX_MAIL_WORDS 512           /* per slot, word 0 = sequence flag: up to 511 payload words (8 bytes each) */
// post: after everything enqueued on `st` so far, the 64-bit words of up to four device arrays travel to the host, one after the other; wait: blocks until
// they are there (payload = the words).  read / gather: post + wait + copy out.
struct MailSrc { const void* p[4]; int n[4]; int k; };
int svx_mail_post(svx_ctx* c, hipStream_t st, const MailSrc& src, unsigned long long* ticket);
int svx_mail_wait(svx_ctx* c, hipStream_t st, unsigned long long ticket, const unsigned long long** payload);
inline int svx_mail_gather(svx_ctx* c, hipStream_t st, int k, const void* const* srcs, const int* n_words, void* const* dsts) {
    MailSrc m; memset(&m, 0, sizeof m);
    m.k = k;
    for (int j = 0; j < k; j++) { m.p[j] = srcs[j]; m.n[j] = n_words[j]; }
    unsigned long long t = 0; const unsigned long long* w = nullptr;
    SVXCHK(svx_mail_post(c, st, m, &t));
    SVXCHK(svx_mail_wait(c, st, t, &w));
    for (int j = 0; j < k; j++) { memcpy(dsts[j], w, (size_t)n_words[j] * 8); w += n_words[j]; }
    return SVX_OK;
}
inline int svx_mail_read(svx_ctx* c, hipStream_t st, const void* dev_src, int n_words, void* host_dst) {
    const void* srcs[1] = {dev_src}; const int n[1] = {n_words}; void* dsts[1] = {host_dst};
    return svx_mail_gather(c, st, 1, srcs, n, dsts);
}
inline int svx_mail_read2(svx_ctx* c, hipStream_t st, const void* a, int na, void* da, const void* b, int nb, void* db) {
    const void* srcs[2] = {a, b}; const int n[2] = {na, nb}; void* dsts[2] = {da, db};
    return svx_mail_gather(c, st, 2, srcs, n, dsts);
}
inline int svx_mail_read3(svx_ctx* c, hipStream_t st, const void* a, int na, void* da, const void* b, int nb, void* db, const void* e, int ne, void* de) {
    const void* srcs[3] = {a, b, e}; const int n[3] = {na, nb, ne}; void* dsts[3] = {da, db, de};
    return svx_mail_gather(c, st, 3, srcs, n, dsts);
}

void svx_preload_collect(); void svx_preload_cluster(); void svx_preload_edit(); void svx_preload_prims();      // code objects loaded at context creation

// ---- stage entry points ------------------------------------------------------------------------------------
int svx_collect_impl(svx_ctx* c, const svx_batch* b_dev, const svx_params* p);
int svx_cluster_impl(svx_ctx* c, const ClusterIn& in, int32_t n_contig, const int32_t* rank_dev, const svx_params* p);
void svx_exchange_poison(svx_ctx* c);
int svx_edit_prepack_begin(svx_ctx* c, const ClusterIn& in, const svx_params& p, hipEvent_t input_ready);
int svx_edit_prepack_pack(svx_ctx* c, const ClusterIn& in);
int svx_pair_distances_impl(svx_ctx* c, const ClusterIn& in, int64_t n_pairs, const int64_t* ia_dev, const int64_t* ib_dev, const svx_params* p, double* out_dev);
int svx_set_alignment_index_impl(svx_ctx* c, const svx_aln_index* h);
int svx_genotype_impl(svx_ctx* c, int32_t mode, int64_t n_cand, const int32_t* tid, const int32_t* start, const int32_t* end, const int64_t* moff,
                      const int32_t* mnames, int32_t min_mapq, int32_t* out);
int svx_edit_distance_pairs(svx_ctx* c, int64_t n_pairs, const uint8_t* codes_dev, const int64_t* a_off_dev, const int64_t* b_off_dev, int32_t* out_dev);
int svx_linkage_batch(svx_ctx* c, int64_t n_problems, const int32_t* n_dev, const int64_t* d_off_dev, const double* d_dev, double cutoff,
                      const int64_t* label_off_dev, int32_t* labels_dev);

// ---- device helpers --------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// 32-bit inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (7 dependent VALU steps; __shfl_up goes through
// ds_bpermute, the LDS crossbar, at ~10x the latency).  A lane switched off by the bank / row mask receives `old` = 0.
#define SVX_DPP_ADD(dst_, src_, ctrl_, row_, bank_) dst_ += __builtin_amdgcn_update_dpp(0, src_, ctrl_, row_, bank_, true)
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    int s = v;
    SVX_DPP_ADD(s, v, 0x111, 0xf, 0xf);          // row_shr:1
    SVX_DPP_ADD(s, v, 0x112, 0xf, 0xf);          // row_shr:2
    SVX_DPP_ADD(s, v, 0x113, 0xf, 0xf);          // row_shr:3   -> 4 neighbours
    SVX_DPP_ADD(s, s, 0x114, 0xf, 0xe);          // row_shr:4, lanes 4..15 of a row  -> 8
    SVX_DPP_ADD(s, s, 0x118, 0xf, 0xc);          // row_shr:8, lanes 8..15           -> the row of 16
    SVX_DPP_ADD(s, s, 0x142, 0xa, 0xf);          // row_bcast:15 into rows 1 and 3
    SVX_DPP_ADD(s, s, 0x143, 0xc, 0xf);          // row_bcast:31 into rows 2 and 3
    return s;
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), 63); }
// minimum of a double over the 64 lanes, the same in every lane afterwards (same DPP ladder; a lane the shift does not reach keeps its own value)
#define SVX_DPP_MIN_F64(s_, src_, ctrl_, row_, bank_)                                                                     \
    {                                                                                                                     \
        const int lo_ = __builtin_amdgcn_update_dpp(__double2loint(src_), __double2loint(src_), ctrl_, row_, bank_, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(__double2hiint(src_), __double2hiint(src_), ctrl_, row_, bank_, false); \
        const double t_ = __hiloint2double(hi_, lo_);                                                                     \
        s_ = t_ < s_ ? t_ : s_;                                                                                           \
    }
__device__ __forceinline__ double wave_min_f64(double v) {
    double s = v;
    SVX_DPP_MIN_F64(s, v, 0x111, 0xf, 0xf);
    SVX_DPP_MIN_F64(s, v, 0x112, 0xf, 0xf);
    SVX_DPP_MIN_F64(s, v, 0x113, 0xf, 0xf);
    { const double u = s; SVX_DPP_MIN_F64(s, u, 0x114, 0xf, 0xe); }
    { const double u = s; SVX_DPP_MIN_F64(s, u, 0x118, 0xf, 0xc); }
    { const double u = s; SVX_DPP_MIN_F64(s, u, 0x142, 0xa, 0xf); }
    { const double u = s; SVX_DPP_MIN_F64(s, u, 0x143, 0xc, 0xf); }
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 63), __builtin_amdgcn_readlane(__double2loint(s), 63));
}
// entry i (wave-uniform, < 128) of a table kept in two registers across the lanes: lane l holds entries l and l + 64
__device__ __forceinline__ int lane_table_get(int v0, int v1, int i) {
    return i < 64 ? __builtin_amdgcn_readlane(v0, i) : __builtin_amdgcn_readlane(v1, i - 64);
}
__device__ __forceinline__ void lane_table_set(int& v0, int& v1, int i, int val) {
    const int lane = (int)(threadIdx.x & 63);
    if (lane == i) v0 = val;
    if (lane + 64 == i) v1 = val;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scan across the 64 lanes
__device__ __forceinline__ long long wave_incl_scan_i64(long long v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t = __shfl_up(v, o, 64);
        if (lane_id() >= o) v += t;
    }
    return v;
}

#endif
