// api.hip - the extern "C" boundary of libsvx.so (include/svx.h).
#include "common.hpp"
#include "hostcopy.hpp"
#include <utility>
#include <cstdlib>

thread_local std::string g_svx_err;

int svx_fail(int code, const char* what, const char* file, int line, hipError_t e) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed at %s:%d%s%s", what, file, line, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    g_svx_err = buf;
    return code;
}

// ---- guard-mode allocator (common.hpp) --------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <unordered_map>
bool svx_guard_mode() { static const bool on = []() { const char* e = getenv("SVX_ALLOC_GUARD"); return e && e[0] == '1'; }(); return on; }
bool svx_guard_slack() { static const bool on = []() { const char* e = getenv("SVX_ALLOC_GUARD_SLACK"); return e && e[0] == '1'; }(); return on; }
namespace {
struct GuardRec { void* va; size_t va_len; void* map_at; size_t map_len; hipMemGenericAllocationHandle_t h; };
std::mutex g_guard_mutex;
std::unordered_map<void*, GuardRec> g_guard;
}
int svx_guard_alloc(void** out, size_t bytes) {
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; HIPCHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (!gran) gran = 4096;
    // the buffer's end sits on the end of the mapping up to the alignment (256 like hipMalloc; SVX_ALLOC_GUARD_ALIGN=16: strict - an overrun of one element faults)
    static const size_t align = []() { const char* e = getenv("SVX_ALLOC_GUARD_ALIGN"); const long long v = e ? atoll(e) : 0; return (size_t)(v >= 16 && (v & (v - 1)) == 0 ? v : 256); }();
    const size_t want = (bytes + align - 1) & ~(align - 1);
    const size_t map_len = ((want ? want : 16) + gran - 1) / gran * gran;
    // (the mapping starts at the start of the reservation: the runtime resolves a pointer inside a mapping against its RESERVATION when it copies or fills, so a
    // mapping at an offset makes every hipMemcpy / hipMemset land one granule off; the unmapped granule is behind the buffer only)
    GuardRec r; r.map_len = map_len; r.va_len = map_len + gran;
    HIPCHK(hipMemAddressReserve(&r.va, r.va_len, gran, nullptr, 0));
    r.map_at = r.va;
    HIPCHK(hipMemCreate(&r.h, map_len, &prop, 0));
    HIPCHK(hipMemMap(r.map_at, map_len, 0, r.h, 0));
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc);
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    HIPCHK(hipMemSetAccess(r.map_at, map_len, &acc, 1));
    void* p = (char*)r.map_at + (map_len - want);
    { std::lock_guard<std::mutex> g(g_guard_mutex); g_guard[p] = r; }
    *out = p;
    return SVX_OK;
}
void svx_guard_free(void* p) {
    GuardRec r;
    { std::lock_guard<std::mutex> g(g_guard_mutex); auto it = g_guard.find(p); if (it == g_guard.end()) { (void)hipFree(p); return; } r = it->second; g_guard.erase(it); }
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(r.map_at, r.map_len);
    (void)hipMemRelease(r.h);
    // the address range is NOT given back (SVX_ALLOC_GUARD_REUSE_VA=1 does): a later reservation at the same address is a different physical allocation, and a
    // stale pointer into the old one should fault, not read the new one; 47 bits of address space outlast any test run
    static const bool reuse = []() { const char* e = getenv("SVX_ALLOC_GUARD_REUSE_VA"); return e && e[0] == '1'; }();
    if (reuse) (void)hipMemAddressFree(r.va, r.va_len);
}

extern "C" const char* svx_last_error(void) { return g_svx_err.c_str(); }
extern "C" int svx_version(void) { return 100; }

extern "C" int svx_ctx_create(int device_ordinal, svx_ctx** out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || device_ordinal >= ndev)
        return svx_fail(SVX_E_NODEVICE, "no HIP device available (libsvx has no CPU fallback)", __FILE__, __LINE__, e);
    HIPCHK(hipSetDevice(device_ordinal));
    svx_ctx* c = new svx_ctx();
    c->device = device_ordinal;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount; }
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& ev : c->ev) HIPCHK(hipEventCreate(&ev));
    {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        // Which kind of edit-distance launch gets the high-priority streams.  The full-matrix launches hold the longest serial chains (a pair of 5000 x 5000
        // symbols is 5000 dependent steps whatever the width) and, since the band windows narrow, most of the work: they go first, the band launches fill in
        // (configs[1]: 16.9 ms against 17.9 the other way round, profiles/r04_edit_prio_ab.txt).  SVX_EDIT_PRIO=band / equal: A/B switch.
        const char* pe = getenv("SVX_EDIT_PRIO");
        const int mode = pe && !strcmp(pe, "band") ? 0 : (pe && !strcmp(pe, "equal") ? 2 : 1);
        for (int k = 0; k < SVX_N_AUX; k++) {
            const bool band_stream = k < 2 || k == 5;
            int prio = mode == 2 ? least : ((band_stream == (mode == 0)) ? greatest : least);
            if (k == 6) prio = (least + greatest) / 2;                 // the main stream's own priority
            HIPCHK(hipStreamCreateWithPriority(&c->aux[k], hipStreamNonBlocking, prio));
        }
    }
    { void* hp = nullptr; HIPCHK(hipHostMalloc(&hp, 4096, hipHostMallocDefault)); c->pinned = (int64_t*)hp; }
    {
        void* hp = nullptr;
        HIPCHK(hipHostMalloc(&hp, (size_t)SVX_MAIL_SLOTS * SVX_MAIL_WORDS * 8, hipHostMallocMapped | hipHostMallocCoherent));
        memset(hp, 0, (size_t)SVX_MAIL_SLOTS * SVX_MAIL_WORDS * 8);
        c->mail = (unsigned long long*)hp;
        const char* e = getenv("SVX_MAILBOX");
        c->mail_mode = e && e[0] == '0' ? 0 : 1;
    }
    memset(&c->stats, 0, sizeof c->stats);
    svx_preload_collect(); svx_preload_cluster(); svx_preload_edit(); svx_preload_prims();       // code objects now, not inside the first call
    { const char* e = getenv("SVX_EDIT_FORCE_FULL"); c->edit_force_full = e && e[0] == '1'; }
    { const char* e = getenv("SVX_EDIT_GUESS"); if (e && atof(e) > 0) { c->edit_guess = (float)atof(e); c->edit_guess_pinned = true; } }
    *out = c;
    return SVX_OK;
}

extern "C" void svx_ctx_destroy(svx_ctx* c) {
    if (getenv("SVX_ALLOC_STATS")) fprintf(stderr, "svx allocations: %lld hipMalloc calls, %.1f MB, %.2f ms\n", svx_alloc_calls(), svx_alloc_bytes() / 1e6, 1e3 * svx_alloc_seconds());
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& b : c->batch_bufs) b.release();
    c->sig.release(); c->bnd.release(); c->raw_sig.release(); c->raw_bnd.release();
    DevBuf* bufs[] = {&c->counters, &c->raw_indel, &c->shard_cnt, &c->rec_geom, &c->seg_geom, &c->seg_ws, &c->tmp0, &c->tmp1, &c->tmp2, &c->tmp3, &c->tmp4, &c->tmp5, &c->sort_tmp, &c->scan_tmp,
                      &c->g_off, &c->g_codes, &c->c_rank, &c->k_hi, &c->k_lo, &c->k_idx, &c->k_hi2, &c->k_lo2, &c->k_idx2, &c->part_flag, &c->part_id,
                      &c->part_start, &c->part_meta, &c->samp_chain, &c->samp_idx, &c->large_list, &c->samp_stream, &c->cell_shards, &c->mt_words, &c->samp_meta, &c->samp_table, &c->samp_runs, &c->pair_off, &c->ed, &c->work, &c->stage, &c->stage_members, &c->labels, &c->e_words, &c->e_off, &c->e_scratch, &c->e_rec, &c->e_hist, &c->e_desc, &c->e_key, &c->e_val, &c->e_slot, &c->e_fail, &c->e_big_list, &c->e_big_state, &c->e_big_off, &c->e_retry[0], &c->e_retry[1], &c->e_retry[2], &c->e_blk[0], &c->e_blk[1], &c->prepack_tmp,
                      &c->clu.type, &c->clu.contig, &c->clu.start, &c->clu.end, &c->clu.contig2, &c->clu.start2, &c->clu.end2, &c->clu.aux, &c->clu.score,
                      &c->clu.std_span, &c->clu.std_pos, &c->clu.size, &c->clu.member_off, &c->clu.members, &c->clu.part_index};
    for (auto* b : bufs) b->release();
    for (auto& b : c->user_sig) b.release();
    for (auto& b : c->geno) b.release();
    for (auto& ev : c->ev) (void)hipEventDestroy(ev);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->mail) (void)hipHostFree(c->mail);
    for (auto& a : c->aux) (void)hipStreamDestroy(a);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- mailbox (common.hpp) -----------------------------------------------------------------------------------------------------------------------------
// One wave copies the words and, after a system-scope fence, the sequence number of the post into word 0 of the slot; the host spins on that word.  Sequence
// numbers only grow, so a slot that is reused can never show an older post as the new one.  The wait keeps an eye on the stream: a kernel that faulted
// never posts, and hipStreamQuery says so.
__global__ __launch_bounds__(64) void k_mail(MailSrc src, unsigned long long* slot, unsigned long long seq) {
    int at = 1;
    for (int j = 0; j < src.k; j++) {
        const unsigned long long* p = (const unsigned long long*)src.p[j];
        for (int i = (int)threadIdx.x; i < src.n[j]; i += 64) __hip_atomic_store(slot + at + i, p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        at += src.n[j];
    }
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) __hip_atomic_store(slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int svx_mail_post(svx_ctx* c, hipStream_t st, const MailSrc& src, unsigned long long* ticket) {
    int total = 0;
    if (src.k < 0 || src.k > 4) return svx_fail(SVX_E_ARG, "svx_mail_post: bad source count", __FILE__, __LINE__, hipSuccess);
    for (int j = 0; j < src.k; j++) { if (src.n[j] < 0) return svx_fail(SVX_E_ARG, "svx_mail_post: negative length", __FILE__, __LINE__, hipSuccess); total += src.n[j]; }
    if (total >= SVX_MAIL_WORDS) return svx_fail(SVX_E_ARG, "svx_mail_post: too many words", __FILE__, __LINE__, hipSuccess);
    const unsigned long long seq = ++c->mail_seq;
    unsigned long long* slot = c->mail + (size_t)(seq % SVX_MAIL_SLOTS) * SVX_MAIL_WORDS;
    if (c->mail_mode) {
        k_mail<<<1, 64, 0, st>>>(src, slot, seq);
        HIPCHK(hipGetLastError());
    } else {
        int at = 1;
        for (int j = 0; j < src.k; j++) {
            if (src.n[j]) HIPCHK(hipMemcpyAsync(slot + at, src.p[j], (size_t)src.n[j] * 8, hipMemcpyDeviceToHost, st));
            at += src.n[j];
        }
    }
    *ticket = seq;
    return SVX_OK;
}

int svx_mail_wait(svx_ctx* c, hipStream_t st, unsigned long long ticket, const unsigned long long** payload) {
    unsigned long long* slot = c->mail + (size_t)(ticket % SVX_MAIL_SLOTS) * SVX_MAIL_WORDS;
    *payload = slot + 1;
    if (!c->mail_mode) { HIPCHK(hipStreamSynchronize(st)); return SVX_OK; }
    for (long long spin = 0;; spin++) {
        if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == ticket) return SVX_OK;
        __builtin_ia32_pause();
        if ((spin & 0xffff) == 0xffff) {                      // every few hundred microseconds: is the stream still alive?
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) {
                if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == ticket) return SVX_OK;
                return svx_fail(SVX_E_STATE, "mailbox: the stream ran dry without the post", __FILE__, __LINE__, hipSuccess);
            }
            if (q != hipErrorNotReady) return svx_fail(SVX_E_HIP, "mailbox: stream failed while a post was awaited", __FILE__, __LINE__, q);
        }
    }
}

extern "C" void* svx_stream(svx_ctx* c) { return (void*)c->stream; }
// test / inspection helper: device memory -> host (a device-resident svx_batch handed out by the BAM reader can be looked at without torch)
extern "C" int svx_memcpy_d2h(void* host_dst, const void* device_src, uint64_t bytes) {
    // (a fault of a kernel launched EARLIER, on any stream of the process, surfaces at the next synchronising call: told apart from a bad copy here)
    { const hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) return svx_fail(SVX_E_HIP, "a kernel launched before this copy faulted (svx_memcpy_d2h only noticed it)", __FILE__, __LINE__, e); }
    return svx_d2h(host_dst, device_src, (size_t)bytes, nullptr);         // through the library's page-locked buffers: the runtime never page-locks the caller's memory
}
// test / inspection helpers beside it: library-owned device memory, a host -> device copy, and "is the device clean" (any kernel fault of the process shows here)
extern "C" void* svx_dev_alloc(uint64_t bytes) {
    void* p = nullptr;
    const hipError_t e = hipMalloc(&p, (size_t)(bytes ? bytes : 1));
    if (e != hipSuccess) { (void)svx_fail(SVX_E_HIP, "hipMalloc", __FILE__, __LINE__, e); return nullptr; }
    return p;
}
extern "C" void svx_dev_free(void* p) { if (p) (void)hipFree(p); }
extern "C" int svx_memcpy_h2d(void* device_dst, const void* host_src, uint64_t bytes) {
    SVXCHK(svx_h2d(device_dst, host_src, (size_t)bytes, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    return SVX_OK;
}
extern "C" int svx_device_synchronize(void) { HIPCHK(hipDeviceSynchronize()); return SVX_OK; }
extern "C" int svx_get_stats(svx_ctx* c, svx_stats* out) { *out = c->stats; return SVX_OK; }

static int upload(svx_ctx* c, DevBuf& d, const void* host, size_t bytes, size_t pad = 64) {
    SVXCHK(d.reserve(bytes + pad));
    return svx_h2d(d.p, host, bytes, c->stream);            // (hostcopy.hpp: `host` has been read when this returns)
}
// the same inside a batch of copies: `host` has been read after hc.finish() (arrays in the library's own page-locked memory are copied in place, hostcopy.hpp)
static int upload(svx_ctx* c, HostCopy& hc, DevBuf& d, const void* host, size_t bytes, size_t pad = 64) {
    (void)c;
    SVXCHK(d.reserve(bytes + pad));
    return hc.h2d(d.p, host, bytes);
}

// ---- COLLECT -----------------------------------------------------------------------------------------------
__global__ void k_acc_fix(long long n, uint64_t* key, uint64_t key_add, int64_t* seq_off_dst, const int64_t* seq_off_src, int64_t seq_add) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] += key_add;
    if (i <= n) seq_off_dst[i] = seq_off_src[i] + seq_add;
}

// append the batch table `s` (sorted by key) to the accumulator: keys shifted by the batch's slot base, sequence offsets by the bases held so far
static int append_sigs(svx_ctx* c, DevSigs& acc, DevSigs& s) {
    hipStream_t st = c->stream;
    const int64_t n = s.n, at = acc.n;
    SVXCHK(acc.reserve_keep(at + n + 1, st));
    SVXCHK(acc.seq.reserve((size_t)(acc.n_seq + s.n_seq) + 16, true, st));
#define APP(f, w) do { if (n) HIPCHK(hipMemcpyAsync(acc.f.as<char>() + (size_t)at * (w), s.f.p, (size_t)n * (w), hipMemcpyDeviceToDevice, st)); } while (0)
    APP(key, 8); APP(type, 1); APP(src, 1); APP(aux, 1); APP(contig, 4); APP(start, 4); APP(end, 4); APP(contig2, 4); APP(pos2, 4); APP(read_id, 4);
#undef APP
    if (s.n_seq) HIPCHK(hipMemcpyAsync(acc.seq.as<char>() + acc.n_seq, s.seq.p, (size_t)s.n_seq, hipMemcpyDeviceToDevice, st));
    k_acc_fix<<<(unsigned)((n + 1 + 255) / 256), 256, 0, st>>>(n, acc.key.as<uint64_t>() + at, c->slot_base << 32, acc.seq_off.as<int64_t>() + at,
                                                              s.seq_off.as<int64_t>(), acc.n_seq);
    HIPCHK(hipGetLastError());
    acc.n = at + n; acc.n_seq += s.n_seq;
    return SVX_OK;
}

extern "C" int svx_collect(svx_ctx* c, const svx_batch* b, const svx_params* p) {
    if (!c || !b || !p) return svx_fail(SVX_E_ARG, "null argument", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    svx_batch d = *b;
    if (!b->on_device) {
        // host batch: read the array extents from the offset tables and stage everything in HBM
        const size_t n = (size_t)b->n_rec, ns = (size_t)b->n_seg;
        const size_t n_ops = n ? (size_t)b->cigar_off[n] : 0, n_seq = n ? (size_t)b->seq_off[n] : 0;
        const size_t n_sops = ns ? (size_t)b->seg_cigar_off[ns] : 0;
        if (c->batch_bufs.size() < 25) c->batch_bufs.resize(25);
        auto& B = c->batch_bufs;
        HostCopy hc(c->stream);
        SVXCHK(upload(c, hc, B[0], b->flag, n * 2)); d.flag = B[0].as<uint16_t>();
        SVXCHK(upload(c, hc, B[1], b->tid, n * 4)); d.tid = B[1].as<int32_t>();
        SVXCHK(upload(c, hc, B[2], b->pos, n * 4)); d.pos = B[2].as<int32_t>();
        SVXCHK(upload(c, hc, B[3], b->mapq, n)); d.mapq = B[3].as<uint8_t>();
        SVXCHK(upload(c, hc, B[4], b->lseq, n * 4)); d.lseq = B[4].as<int32_t>();
        SVXCHK(upload(c, hc, B[5], b->read_id, n * 4)); d.read_id = B[5].as<int32_t>();
        SVXCHK(upload(c, hc, B[6], b->order, n * 4)); d.order = B[6].as<uint32_t>();
        SVXCHK(upload(c, hc, B[7], b->seg_order, n * 4)); d.seg_order = B[7].as<uint32_t>();
        SVXCHK(upload(c, hc, B[8], b->cigar_off, (n + 1) * 8)); d.cigar_off = B[8].as<uint64_t>();
        SVXCHK(upload(c, hc, B[9], b->cigar, n_ops * 4)); d.cigar = B[9].as<uint32_t>();
        SVXCHK(upload(c, hc, B[10], b->seq_off, (n + 1) * 8)); d.seq_off = B[10].as<uint64_t>();
        SVXCHK(upload(c, hc, B[11], b->seq, n_seq)); d.seq = B[11].as<uint8_t>();
        SVXCHK(upload(c, hc, B[12], b->seg_off, (n + 1) * 4)); d.seg_off = B[12].as<uint32_t>();
        SVXCHK(upload(c, hc, B[13], b->seg_tid, ns * 4)); d.seg_tid = B[13].as<int32_t>();
        SVXCHK(upload(c, hc, B[14], b->seg_pos, ns * 4)); d.seg_pos = B[14].as<int32_t>();
        SVXCHK(upload(c, hc, B[15], b->seg_rev, ns)); d.seg_rev = B[15].as<uint8_t>();
        SVXCHK(upload(c, hc, B[16], b->seg_mapq, ns)); d.seg_mapq = B[16].as<uint8_t>();
        SVXCHK(upload(c, hc, B[17], b->seg_lseq, ns * 4)); d.seg_lseq = B[17].as<int32_t>();
        SVXCHK(upload(c, hc, B[18], b->seg_cigar_off, (ns + 1) * 8)); d.seg_cigar_off = B[18].as<uint64_t>();
        SVXCHK(upload(c, hc, B[19], b->seg_cigar, n_sops * 4)); d.seg_cigar = B[19].as<uint32_t>();
        SVXCHK(upload(c, hc, B[20], b->contig_rank, (size_t)b->n_contig * 4)); d.contig_rank = B[20].as<int32_t>();
        if (b->seq_rng_off) {
            const size_t nr = (size_t)b->n_seq_rng;
            SVXCHK(upload(c, hc, B[21], b->seq_rng_off, (n + 1) * 4)); d.seq_rng_off = B[21].as<uint32_t>();
            SVXCHK(upload(c, hc, B[22], b->seq_rng_q0, nr * 4)); d.seq_rng_q0 = B[22].as<int32_t>();
            SVXCHK(upload(c, hc, B[23], b->seq_rng_len, nr * 4)); d.seq_rng_len = B[23].as<int32_t>();
            SVXCHK(upload(c, hc, B[24], b->seq_rng_byte, nr * 8)); d.seq_rng_byte = B[24].as<uint64_t>();
        }
        SVXCHK(hc.finish());
        d.on_device = 1;
    }
    SVXCHK(svx_collect_impl(c, &d, p));
    if (c->accumulate) {
        SVXCHK(append_sigs(c, c->acc_sig, c->sig));
        SVXCHK(append_sigs(c, c->acc_bnd, c->bnd));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SVX_OK;
}

// One input file usually arrives as several record batches (svx_bam_read_batch): with accumulation on, every svx_collect APPENDS its
// two signature lists to the lists resident in the context - keys are made global by `slot_base` (the caller passes the number of
// emission slots of all earlier batches, 2 per record is always enough) - so that svx_cluster(source 0 / 1) sees the whole file without
// the tables ever leaving HBM.  mode 1 starts a fresh accumulation, 0 returns to one-batch-per-call semantics.
extern "C" int svx_collect_accumulate(svx_ctx* c, int mode) {
    if (!c) return svx_fail(SVX_E_ARG, "null context", __FILE__, __LINE__, hipSuccess);
    if (c->accumulate && mode == 0) {
        // the accumulated lists become "the result of the last COLLECT": svx_collect_count / _fetch / svx_cluster(source 0 / 1) keep seeing
        // the whole file after accumulation is switched off (buffers are exchanged, nothing is copied)
        auto promote = [](DevSigs& s, DevSigs& a) {
            std::swap(s.key, a.key); std::swap(s.type, a.type); std::swap(s.src, a.src); std::swap(s.aux, a.aux); std::swap(s.contig, a.contig);
            std::swap(s.start, a.start); std::swap(s.end, a.end); std::swap(s.contig2, a.contig2); std::swap(s.pos2, a.pos2); std::swap(s.read_id, a.read_id);
            std::swap(s.seq_off, a.seq_off); std::swap(s.seq, a.seq);
            const int64_t cap = s.cap < a.cap ? s.cap : a.cap;         // every buffer of either table holds at least this many rows
            s.n = a.n; s.n_seq = a.n_seq; s.cap = cap; a.cap = cap;
        };
        promote(c->sig, c->acc_sig); promote(c->bnd, c->acc_bnd);
    }
    c->accumulate = mode != 0;
    c->acc_sig.n = c->acc_sig.n_seq = 0; c->acc_bnd.n = c->acc_bnd.n_seq = 0;
    c->slot_base = 0;
    return SVX_OK;
}
extern "C" int svx_collect_set_slot_base(svx_ctx* c, uint64_t slot_base) {
    if (!c) return svx_fail(SVX_E_ARG, "null context", __FILE__, __LINE__, hipSuccess);
    c->slot_base = slot_base;
    return SVX_OK;
}

extern "C" int svx_collect_count(svx_ctx* c, int64_t* n_sig, int64_t* n_seq, int64_t* n_bnd) {
    const DevSigs& s = c->accumulate ? c->acc_sig : c->sig;
    const DevSigs& b = c->accumulate ? c->acc_bnd : c->bnd;
    if (n_sig) *n_sig = s.n;
    if (n_seq) *n_seq = s.n_seq;
    if (n_bnd) *n_bnd = b.n;
    return SVX_OK;
}

extern "C" int svx_collect_fetch(svx_ctx* c, int which, svx_sig_view* o) {
    HIPCHK(hipSetDevice(c->device));
    DevSigs& s = c->accumulate ? (which ? c->acc_bnd : c->acc_sig) : (which ? c->bnd : c->sig);
    const size_t n = (size_t)s.n;
    hipStream_t st = c->stream;
    // o->on_device: the caller's arrays live in HBM (e.g. torch tensors feeding an RCCL all-gather) -> device-to-device copies
    HostCopy hc(st);
    const bool dev = o->on_device != 0;
#define CPY(dst, buf, bytes) do { if ((bytes) && (dst)) { if (dev) HIPCHK(hipMemcpyAsync((dst), (buf).p, (bytes), hipMemcpyDeviceToDevice, st)); else SVXCHK(hc.d2h((dst), (buf).p, (bytes))); } } while (0)
    CPY(o->key, s.key, n * 8); CPY(o->type, s.type, n); CPY(o->src, s.src, n); CPY(o->aux, s.aux, n);
    CPY(o->contig, s.contig, n * 4); CPY(o->start, s.start, n * 4); CPY(o->end, s.end, n * 4); CPY(o->contig2, s.contig2, n * 4);
    CPY(o->pos2, s.pos2, n * 4); CPY(o->read_id, s.read_id, n * 4);
    CPY(o->seq_off, s.seq_off, (n + 1) * 8);
    CPY(o->seq, s.seq, (size_t)s.n_seq);
#undef CPY
    SVXCHK(hc.finish());
    HIPCHK(hipStreamSynchronize(st));
    o->n = s.n;
    return SVX_OK;
}

// ---- genome --------------------------------------------------------------------------------------------------
extern "C" int svx_set_genome(svx_ctx* c, const svx_genome* g) {
    HIPCHK(hipSetDevice(c->device));
    c->g_n = g->n_contig;
    if (g->on_device) {
        c->g_off_p = g->off; c->g_codes_p = g->codes; c->g_borrowed = true;
        return SVX_OK;
    }
    const size_t n = (size_t)g->n_contig;
    const size_t tot = (size_t)g->off[n];
    SVXCHK(upload(c, c->g_off, g->off, (n + 1) * 8));
    SVXCHK(upload(c, c->g_codes, g->codes, tot));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->g_off_p = c->g_off.as<int64_t>(); c->g_codes_p = c->g_codes.as<uint8_t>(); c->g_borrowed = false;
    return SVX_OK;
}

// ---- CLUSTER -------------------------------------------------------------------------------------------------
extern "C" int svx_cluster(svx_ctx* c, int source, const svx_sig_view* sigs, int32_t n_contig, const int32_t* contig_rank_host, const svx_params* p) {
    if (!c || !p || !contig_rank_host) return svx_fail(SVX_E_ARG, "null argument", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    SVXCHK(upload(c, c->c_rank, contig_rank_host, (size_t)n_contig * 4));
    ClusterIn in;
    if (source == 0 || source == 1) {
        DevSigs& s = c->accumulate ? (source ? c->acc_bnd : c->acc_sig) : (source ? c->bnd : c->sig);
        in.n = s.n; in.type = s.type.as<uint8_t>(); in.aux = s.aux.as<uint8_t>(); in.contig = s.contig.as<int32_t>(); in.start = s.start.as<int32_t>();
        in.end = s.end.as<int32_t>(); in.contig2 = s.contig2.as<int32_t>(); in.pos2 = s.pos2.as<int32_t>(); in.read_id = s.read_id.as<int32_t>();
        in.seq_off = s.seq_off.as<int64_t>(); in.seq = s.seq.as<uint8_t>();
        if (s.n > 0 && !s.seq_off.p) return svx_fail(SVX_E_STATE, "no resident signatures: run svx_collect first", __FILE__, __LINE__, hipSuccess);
    } else if (source == 2) {
        if (!sigs) return svx_fail(SVX_E_ARG, "source 2 needs a signature table", __FILE__, __LINE__, hipSuccess);
        in.n = sigs->n;
        if (sigs->on_device) {
            in.type = sigs->type; in.aux = sigs->aux; in.contig = sigs->contig; in.start = sigs->start; in.end = sigs->end; in.contig2 = sigs->contig2;
            in.pos2 = sigs->pos2; in.read_id = sigs->read_id; in.seq_off = sigs->seq_off; in.seq = sigs->seq;
        } else {
            const size_t n = (size_t)sigs->n;
            const size_t nseq = n ? (size_t)sigs->seq_off[n] : 0;
            auto& U = c->user_sig;
            SVXCHK(upload(c, U[0], sigs->type, n)); in.type = U[0].as<uint8_t>();
            SVXCHK(upload(c, U[1], sigs->aux, n)); in.aux = U[1].as<uint8_t>();
            SVXCHK(upload(c, U[2], sigs->contig, n * 4)); in.contig = U[2].as<int32_t>();
            SVXCHK(upload(c, U[3], sigs->start, n * 4)); in.start = U[3].as<int32_t>();
            SVXCHK(upload(c, U[4], sigs->end, n * 4)); in.end = U[4].as<int32_t>();
            SVXCHK(upload(c, U[5], sigs->contig2, n * 4)); in.contig2 = U[5].as<int32_t>();
            SVXCHK(upload(c, U[6], sigs->pos2, n * 4)); in.pos2 = U[6].as<int32_t>();
            SVXCHK(upload(c, U[7], sigs->read_id, n * 4)); in.read_id = U[7].as<int32_t>();
            SVXCHK(upload(c, U[8], sigs->seq_off, (n + 1) * 8)); in.seq_off = U[8].as<int64_t>();
            SVXCHK(upload(c, U[9], sigs->seq, nseq)); in.seq = U[9].as<uint8_t>();
        }
    } else return svx_fail(SVX_E_ARG, "source must be 0, 1 or 2", __FILE__, __LINE__, hipSuccess);
    return svx_cluster_impl(c, in, n_contig, c->c_rank.as<int32_t>(), p);
}

// Debug / test hook: span_position_distance (src/svim/SVIM_clustering.py:47-96) of n_pairs signature pairs (a[k], b[k]) of a HOST table,
// computed by the same device code the clustering runs (FP64 arithmetic and haplotype edit distances included).
extern "C" int svx_pair_distances(svx_ctx* c, const svx_sig_view* sigs, int64_t n_pairs, const int64_t* a_host, const int64_t* b_host, const svx_params* p,
                                  double* out_host) {
    if (!c || !sigs || !p || sigs->on_device) return svx_fail(SVX_E_ARG, "svx_pair_distances wants a host table", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    ClusterIn in;
    const size_t n = (size_t)sigs->n;
    const size_t nseq = n ? (size_t)sigs->seq_off[n] : 0;
    auto& U = c->user_sig;
    in.n = sigs->n;
    SVXCHK(upload(c, U[0], sigs->type, n)); in.type = U[0].as<uint8_t>();
    SVXCHK(upload(c, U[1], sigs->aux, n)); in.aux = U[1].as<uint8_t>();
    SVXCHK(upload(c, U[2], sigs->contig, n * 4)); in.contig = U[2].as<int32_t>();
    SVXCHK(upload(c, U[3], sigs->start, n * 4)); in.start = U[3].as<int32_t>();
    SVXCHK(upload(c, U[4], sigs->end, n * 4)); in.end = U[4].as<int32_t>();
    SVXCHK(upload(c, U[5], sigs->contig2, n * 4)); in.contig2 = U[5].as<int32_t>();
    SVXCHK(upload(c, U[6], sigs->pos2, n * 4)); in.pos2 = U[6].as<int32_t>();
    SVXCHK(upload(c, U[7], sigs->read_id, n * 4)); in.read_id = U[7].as<int32_t>();
    SVXCHK(upload(c, U[8], sigs->seq_off, (n + 1) * 8)); in.seq_off = U[8].as<int64_t>();
    SVXCHK(upload(c, U[9], sigs->seq, nseq)); in.seq = U[9].as<uint8_t>();
    SVXCHK(upload(c, U[10], a_host, (size_t)n_pairs * 8));
    SVXCHK(upload(c, U[11], b_host, (size_t)n_pairs * 8));
    SVXCHK(c->tmp5.reserve((size_t)n_pairs * 8 + 8));
    SVXCHK(svx_pair_distances_impl(c, in, n_pairs, U[10].as<int64_t>(), U[11].as<int64_t>(), p, c->tmp5.as<double>()));
    HIPCHK(hipStreamSynchronize(c->stream));
    return svx_d2h(out_host, c->tmp5.p, (size_t)n_pairs * 8, c->stream);
}

// contig-sharded ranks: this context is rank `rank` of `world`; `fn` all-gathers small host buffers among them (NULL / world <= 1: single rank)
extern "C" int svx_cluster_set_ranks(svx_ctx* c, int rank, int world, svx_allgather_fn fn, void* user) {
    if (!c || world < 1 || rank < 0 || rank >= world) return svx_fail(SVX_E_ARG, "bad rank / world", __FILE__, __LINE__, hipSuccess);
    c->xr_rank = rank; c->xr_world = world; c->xr_fn = fn; c->xr_user = user; c->xr_pending = false;
    return SVX_OK;
}
extern "C" int svx_cluster_stream_positions(svx_ctx* c, int64_t* start, int64_t* end) {
    if (!c) return svx_fail(SVX_E_ARG, "null context", __FILE__, __LINE__, hipSuccess);
    for (int t = 0; t < SVX_NTYPES; t++) { if (start) start[t] = c->stream_start[t]; if (end) end[t] = c->stream_end[t]; }
    return SVX_OK;
}
// a rank that cannot reach its svx_cluster (an error before the call) tells the others, who would otherwise wait for it in the rank exchange
extern "C" int svx_cluster_abort_ranks(svx_ctx* c) {
    if (!c) return svx_fail(SVX_E_ARG, "null context", __FILE__, __LINE__, hipSuccess);
    c->xr_pending = c->xr_world > 1 && c->xr_fn;
    svx_exchange_poison(c);
    return SVX_OK;
}

extern "C" int svx_cluster_count(svx_ctx* c, int64_t* n_clusters, int64_t* n_members) {
    if (n_clusters) *n_clusters = c->clu.n;
    if (n_members) *n_members = c->clu.n_members;
    return SVX_OK;
}

extern "C" int svx_cluster_fetch(svx_ctx* c, svx_cluster_view* o) {
    HIPCHK(hipSetDevice(c->device));
    DevClusters& v = c->clu;
    const size_t n = (size_t)v.n;
    hipStream_t st = c->stream;
    // the destination arrays may be host or device memory (the multi-GPU exchange keeps them in HBM): HostCopy::out looks each one up
    HostCopy hc(st);
#define D2H(dst, buf, bytes) do { if ((bytes) && (dst)) SVXCHK(hc.out((dst), (buf).p, (bytes))); } while (0)
    D2H(o->type, v.type, n); D2H(o->aux, v.aux, n); D2H(o->contig, v.contig, n * 4); D2H(o->start, v.start, n * 4); D2H(o->end, v.end, n * 4);
    D2H(o->contig2, v.contig2, n * 4); D2H(o->start2, v.start2, n * 4); D2H(o->end2, v.end2, n * 4); D2H(o->score, v.score, n * 8);
    D2H(o->std_span, v.std_span, n * 8); D2H(o->std_pos, v.std_pos, n * 8); D2H(o->size, v.size, n * 4);
    if (o->member_off) {
        if (n) D2H(o->member_off, v.member_off, (n + 1) * 8);
        else if (svx_is_device_pointer(o->member_off)) HIPCHK(hipMemsetAsync(o->member_off, 0, 8, st));
        else o->member_off[0] = 0;
    }
    D2H(o->members, v.members, (size_t)v.n_members * 4);
#undef D2H
    SVXCHK(hc.finish());
    HIPCHK(hipStreamSynchronize(st));
    o->n = v.n; o->n_members = v.n_members;
    for (int t = 0; t < SVX_NTYPES; t++) o->type_count[t] = v.type_count[t];
    return SVX_OK;
}

// The partitions of the last svx_cluster (form_partitions, src/svim/SVIM_clustering.py:17-29): `sorted_index` = the signature indices in the order the
// partitions were formed in (type, contig rank(s), coordinate; stable), `part_start[k]` = first position of partition k in it, part_start[n_part] = n_sig.
// A test / inspection hook: the clustering itself never leaves the device.  NULL arrays: counts only.
extern "C" int svx_cluster_partitions_fetch(svx_ctx* c, int64_t* n_sig, int64_t* n_part, uint32_t* sorted_index, int64_t* part_start) {
    if (!c) return svx_fail(SVX_E_ARG, "null context", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->last_cluster_source_n, np = c->stats.n_partitions;
    if (n_sig) *n_sig = n;
    if (n_part) *n_part = np;
    if (n <= 0 || (!sorted_index && !part_start)) return SVX_OK;
    if (!c->k_idx.p || !c->part_start.p) return svx_fail(SVX_E_STATE, "no partitions: run svx_cluster first", __FILE__, __LINE__, hipSuccess);
    HostCopy hc(c->stream);
    if (sorted_index) SVXCHK(hc.d2h(sorted_index, c->k_idx.p, (size_t)n * 4));
    if (part_start) SVXCHK(hc.d2h(part_start, c->part_start.p, (size_t)(np + 1) * 8));
    return hc.finish();
}

// ---- single-function entry points ------------------------------------------------------------------------------
extern "C" int svx_cigar_indel(svx_ctx* c, const uint32_t* cigar_host, int64_t n_ops, int32_t min_length, int64_t* out_pos_ref,
                               int64_t* out_pos_read, int32_t* out_len, uint8_t* out_is_del, int64_t* out_n) {
    // analyze_cigar_indel on ONE alignment: a one-record batch through the same scan kernel (positions relative to the
    // alignment start; sequence length "infinite" so that pos_read comes back unclipped)
    HIPCHK(hipSetDevice(c->device));
    uint16_t flag = 0; int32_t tid = 0, pos = 0, lseq = 0x7fffffff, rid = 0, rank0 = 0; uint8_t mapq = 255; uint32_t order = 0, sorder = 1, seg_off[2] = {0, 0};
    uint64_t coff[2] = {0, (uint64_t)n_ops}, soff[2] = {0, 0}, scoff[1] = {0};
    uint8_t seq = 0; uint32_t zero = 0;
    svx_batch b; memset(&b, 0, sizeof b);
    b.on_device = 0; b.n_rec = 1; b.flag = &flag; b.tid = &tid; b.pos = &pos; b.mapq = &mapq; b.lseq = &lseq; b.read_id = &rid; b.order = &order;
    b.seg_order = &sorder; b.cigar_off = coff; b.cigar = n_ops ? cigar_host : &zero; b.seq_off = soff; b.seq = &seq; b.seg_off = seg_off; b.n_seg = 0;
    b.seg_tid = &tid; b.seg_pos = &pos; b.seg_rev = &seq; b.seg_mapq = &seq; b.seg_lseq = &lseq; b.seg_cigar_off = scoff; b.seg_cigar = &zero;
    b.n_contig = 1; b.contig_rank = &rank0;
    svx_params p; memset(&p, 0, sizeof p);
    p.min_mapq = 0; p.min_sv_size = min_length; p.max_sv_size = 0x7fffffff; p.partition_max_distance = 1000; p.position_distance_normalizer = 900;
    p.edit_distance_normalizer = 1; p.cluster_max_distance = 0.5;
    // sequences are not wanted here: keep the gather from touching `seq` by clearing lseq afterwards is not possible, so run the
    // scan with lseq = INT_MAX and ignore the gathered bases (qlen is clipped only by that bound; seq_off[1] = 0 keeps reads in range)
    c->no_seq_gather = true;
    const int rc = svx_collect(c, &b, &p);
    c->no_seq_gather = false;
    SVXCHK(rc);
    const int64_t n = c->sig.n;
    std::vector<int32_t> start((size_t)n + 1), qpos((size_t)n + 1), end((size_t)n + 1); std::vector<uint8_t> type((size_t)n + 1);
    if (n) {
        HostCopy hc(c->stream);
        SVXCHK(hc.d2h(start.data(), c->sig.start.p, (size_t)n * 4)); SVXCHK(hc.d2h(end.data(), c->sig.end.p, (size_t)n * 4));
        SVXCHK(hc.d2h(qpos.data(), c->sig.qpos.p, (size_t)n * 4)); SVXCHK(hc.d2h(type.data(), c->sig.type.p, (size_t)n));
        SVXCHK(hc.finish());
    }
    for (int64_t i = 0; i < n; i++) {
        out_pos_ref[i] = start[i]; out_len[i] = end[i] - start[i]; out_is_del[i] = type[i] == SVX_DEL;
        out_pos_read[i] = qpos[i];
    }
    *out_n = n;
    return SVX_OK;
}

extern "C" int svx_edit_distance(svx_ctx* c, int64_t n_pairs, const uint8_t* codes_host, const int64_t* a_off, const int64_t* b_off, int32_t* out_dist) {
    HIPCHK(hipSetDevice(c->device));
    if (n_pairs <= 0) return SVX_OK;
    int64_t tot = 0;
    for (int64_t i = 0; i <= n_pairs; i++) { if (a_off[i] > tot) tot = a_off[i]; if (b_off[i] > tot) tot = b_off[i]; }
    SVXCHK(upload(c, c->tmp0, codes_host, (size_t)tot));
    SVXCHK(upload(c, c->tmp1, a_off, (size_t)(n_pairs + 1) * 8));
    SVXCHK(upload(c, c->tmp2, b_off, (size_t)(n_pairs + 1) * 8));
    SVXCHK(c->tmp3.reserve((size_t)n_pairs * 4));
    SVXCHK(svx_edit_distance_pairs(c, n_pairs, c->tmp0.as<uint8_t>(), c->tmp1.as<int64_t>(), c->tmp2.as<int64_t>(), c->tmp3.as<int32_t>()));
    SVXCHK(svx_d2h(out_dist, c->tmp3.p, (size_t)n_pairs * 4, c->stream));
    return SVX_OK;
}

extern "C" int svx_linkage_fcluster(svx_ctx* c, int64_t n_problems, const int32_t* n_host, const int64_t* d_off, const double* d_host, double cutoff,
                                    const int64_t* label_off, int32_t* labels_out) {
    HIPCHK(hipSetDevice(c->device));
    if (n_problems <= 0) return SVX_OK;
    for (int64_t i = 0; i < n_problems; i++)
        if (n_host[i] < 1 || n_host[i] > 100) return svx_fail(SVX_E_ARG, "linkage problems must have 1..100 observations", __FILE__, __LINE__, hipSuccess);
    const int64_t nd = d_off[n_problems], nl = label_off[n_problems];
    SVXCHK(upload(c, c->tmp0, n_host, (size_t)n_problems * 4));
    SVXCHK(upload(c, c->tmp1, d_off, (size_t)(n_problems + 1) * 8));
    SVXCHK(upload(c, c->tmp2, d_host, (size_t)nd * 8));
    SVXCHK(upload(c, c->tmp3, label_off, (size_t)(n_problems + 1) * 8));
    SVXCHK(c->tmp4.reserve((size_t)nl * 4 + 16));
    SVXCHK(svx_linkage_batch(c, n_problems, c->tmp0.as<int32_t>(), c->tmp1.as<int64_t>(), c->tmp2.as<double>(), cutoff, c->tmp3.as<int64_t>(),
                             c->tmp4.as<int32_t>()));
    return svx_d2h(labels_out, c->tmp4.p, (size_t)nl * 4, c->stream);
}

// ---- GENOTYPE ---------------------------------------------------------------------------------------------------------------
extern "C" int svx_set_alignment_index(svx_ctx* c, const svx_aln_index* h) {
    if (!c || !h || h->n < 0 || h->n_contig < 0) return svx_fail(SVX_E_ARG, "bad alignment index", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    return svx_set_alignment_index_impl(c, h);
}
extern "C" int svx_genotype(svx_ctx* c, int32_t mode, int64_t n_cand, const int32_t* cand_tid, const int32_t* cand_start, const int32_t* cand_end,
                            const int64_t* member_off, const int32_t* member_names, int32_t min_mapq, int32_t* out_ref_reads) {
    if (!c || n_cand < 0 || (n_cand && (!cand_tid || !cand_start || !cand_end || !member_off || !out_ref_reads)))
        return svx_fail(SVX_E_ARG, "null argument", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    return svx_genotype_impl(c, mode, n_cand, cand_tid, cand_start, cand_end, member_off, member_names, min_mapq, out_ref_reads);
}
