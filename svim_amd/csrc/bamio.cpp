// bamio.cpp - native BAM front-end: BGZF inflate (multi-threaded) + BAM record decode straight into the
// Structure-of-Arrays record batch of include/svx.h (host memory).
//
// Replaces, for BAM inputs, what the reference gets from pysam/htslib on the COLLECT path:
//   pysam.AlignmentFile(bam).fetch(until_eof=True)        src/svim/SVIM_COLLECT.py:133, src/svim/svim:91
//   record accessors + SA tag string                      src/svim/SVIM_COLLECT.py:47-85,143-145
//   bam_iterator query-name grouping                      src/svim/SVIM_COLLECT.py:8-41,108,113
// Host code (string / inflate work); the numeric work on the arrays built here happens on the GPU (collect.hip).
// SURVEY.md section 8(f) row 1.
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <exception>
#include <chrono>
#include <cstdlib>
#include <future>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <condition_variable>
#include <functional>
#include <mutex>
#include "../../include/svx.h"
#include "devdec.hpp"

extern thread_local std::string g_svx_err;
static int bam_fail(int code, const std::string& what) { g_svx_err = what; return code; }
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// growable array WITHOUT value-initialisation: the big arrays (inflated bytes, CIGAR words, packed bases) are written exactly once by
// the decoder threads; std::vector::resize would first zero them (a memset of every batch)
template <class T> struct RawVec {
    T* p = nullptr; size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec&) = delete; RawVec& operator=(const RawVec&) = delete;
    ~RawVec() { free(p); }
    // Large buffers (inflate windows, CIGAR words, packed bases: hundreds of MB that 100+ threads write for the first time at once) are
    // 2 MB aligned and marked for transparent huge pages: first-touch page faults on 4 KB pages serialise the threads in the kernel.
    void reserve(size_t c) {
        if (c <= cap) return;
        size_t nc = cap + cap / 2 + 1024;
        if (nc < c) nc = c;
        const size_t bytes = nc * sizeof(T);
        if (bytes >= ((size_t)32 << 20)) {
            const size_t huge = (size_t)2 << 20, rounded = (bytes + huge - 1) / huge * huge;
            void* q = nullptr;
            if (posix_memalign(&q, huge, rounded) != 0 || !q) throw std::string("out of host memory");
            (void)madvise(q, rounded, MADV_HUGEPAGE);
            if (p && n) memcpy(q, p, (n < cap ? n : cap) * sizeof(T));
            free(p);
            p = (T*)q; cap = rounded / sizeof(T);
            return;
        }
        T* q = (T*)realloc(p, bytes);
        if (!q) throw std::string("out of host memory");
        p = q; cap = nc;
    }
    void resize_uninit(size_t m) { reserve(m); n = m; }
    void push_back(const T& v) { reserve(n + 1); p[n++] = v; }
    void append(const T* src, size_t m) { reserve(n + m); if (m) memcpy(p + n, src, m * sizeof(T)); n += m; }
    void clear() { n = 0; }
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// Persistent worker threads: a chunk of BGZF blocks inflates in about a millisecond on a many-core host, so spawning threads per chunk
// would cost as much as the work.  run(n, fn) executes fn(0..n-1) on the pool (the caller takes part) and returns when all are done.
struct Pool {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv_go, cv_done;
    std::function<void(int)> job; int n_tasks = 0, next = 0, running = 0; unsigned long long gen = 0; bool stop = false;
    explicit Pool(int n) {
        for (int i = 0; i < n; i++) th.emplace_back([this]() {
            unsigned long long seen = 0;
            for (;;) {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&]() { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                while (next < n_tasks) { const int t = next++; lk.unlock(); job(t); lk.lock(); }
                if (--running == 0) cv_done.notify_all();
            }
        });
    }
    ~Pool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv_go.notify_all(); for (auto& t : th) t.join(); }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th.empty() || n == 1) { for (int t = 0; t < n; t++) fn(t); return; }
        std::unique_lock<std::mutex> lk(m);
        job = fn; n_tasks = n; next = 0; running = (int)th.size() + 1; gen++;
        cv_go.notify_all();
        while (next < n_tasks) { const int t = next++; lk.unlock(); fn(t); lk.lock(); }
        if (--running != 0) cv_done.wait(lk, [&]() { return running == 0; });
    }
};

// the arrays of one record batch; the reader owns TWO sets and alternates between them, so that the batch handed out by
// svx_bam_read_batch stays valid while the next one is being read (the caller overlaps its upload / COLLECT with the next read)
struct BatchArrays {
    std::vector<uint16_t> flag; std::vector<int32_t> tid, bpos, lseq, read_id; std::vector<uint8_t> mapq;
    std::vector<uint32_t> order, seg_order, seg_off, seg_cigar;
    RawVec<uint32_t> cigar; RawVec<uint8_t> seq;
    std::vector<uint64_t> cigar_off, seq_off, seg_cigar_off;
    std::vector<uint8_t> seg_rev, seg_mapq;
    std::vector<int32_t> seg_tid, seg_pos, seg_lseq;
    // sparse SEQ (svx_bam_set_seq_filter): the stored ranges of every record
    std::vector<uint32_t> rng_off; std::vector<int32_t> rng_q0, rng_len; std::vector<uint64_t> rng_byte;
    // per-record SA strings of the current batch (offset, length into sa_blob; length 0 = none)
    std::string sa_blob; std::vector<uint64_t> sa_at; std::vector<uint32_t> sa_len;
};

// Read names -> dense ids in first-seen order.  Interning is the serial part of the reader (ids must follow record order), so it is made
// short: the 64-bit hashes are computed in the parallel decode phase, the table is open addressing over (hash, id) pairs with the slot of
// a later record prefetched, and a hit is confirmed against the name blob (ids are exact, the hash only finds the slot).
static inline uint64_t name_hash(const char* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0xff51afd7ed558ccdull; h ^= h >> 32; p += 8; n -= 8; }
    uint64_t w = 0; memcpy(&w, p, n);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29;
    return h | 1ull;                                       // 0 marks an empty slot
}
struct NameTable {
    struct Slot { uint64_t hash; int32_t id; int32_t len; };
    std::vector<Slot> slots; size_t mask = 0, used = 0;
    std::string blob;                       // NUL-separated, id order
    std::vector<uint64_t> off;              // start of name id in blob
    void grow() {
        const size_t cap = slots.empty() ? (size_t)1 << 16 : slots.size() * 2;
        std::vector<Slot> old; old.swap(slots);
        slots.assign(cap, Slot{0, 0, 0}); mask = cap - 1;
        for (const Slot& e : old) if (e.hash) { size_t i = (size_t)e.hash & mask; while (slots[i].hash) i = (i + 1) & mask; slots[i] = e; }
    }
    void prefetch(uint64_t hash) const { if (mask) __builtin_prefetch(&slots[(size_t)hash & mask]); }
    int32_t find(const char* p, size_t n, uint64_t hash) const {
        if (!mask) return -1;
        for (size_t i = (size_t)hash & mask;; i = (i + 1) & mask) {
            const Slot& e = slots[i];
            if (!e.hash) return -1;
            if (e.hash == hash && (size_t)e.len == n && memcmp(blob.data() + off[(size_t)e.id], p, n) == 0) return e.id;
        }
    }
    int32_t intern(const char* p, size_t n, uint64_t hash) {
        if ((used + 1) * 10 > slots.size() * 6) grow();
        size_t i = (size_t)hash & mask;
        for (;; i = (i + 1) & mask) {
            const Slot& e = slots[i];
            if (!e.hash) break;
            if (e.hash == hash && (size_t)e.len == n && memcmp(blob.data() + off[(size_t)e.id], p, n) == 0) return e.id;
        }
        const int32_t id = (int32_t)off.size();
        off.push_back(blob.size());
        blob.append(p, n); blob.push_back('\0');
        slots[i] = Slot{hash, id, (int32_t)n};
        used++;
        return id;
    }
    size_t size() const { return off.size(); }
};

struct svx_bam {
    // the file is memory-mapped: finding the BGZF block boundaries is a serial walk over 18-byte headers, the compressed payload is read
    // (page-faulted in) by the inflating threads themselves - a serial fread of every block capped the reader at ~2 GB/s of BAM
    int fd = -1; const uint8_t* map = nullptr; size_t map_len = 0, fpos = 0;
    std::string path, sort_order, names_blob;
    std::vector<std::string> ref_names;
    std::vector<int32_t> ref_len, contig_rank;
    std::unordered_map<std::string, int32_t> tid_of;
    NameTable names;                      // read names -> ids
    // uncompressed stream window + the chunk a background thread is inflating meanwhile
    RawVec<uint8_t> buf; size_t pos = 0; bool file_eof = false;
    // The background thread inflates the next chunk into `next` BEHIND `WIN_HEAD` bytes of headroom: switching windows then only moves
    // the unconsumed tail of the old one (a partial record) in front of the new data instead of copying the chunk.
    std::future<void> prefetch; bool prefetch_active = false; RawVec<uint8_t> next; size_t next_len = 0; bool next_eof = false; std::string prefetch_err;
    int n_threads = 8;
    size_t win_head = (size_t)8 << 20, chunk_blocks = 1024, chunk_bytes = (size_t)48 << 20;      // test hooks: SVX_BAM_WIN_HEAD (bytes), SVX_BAM_CHUNK_BLOCKS
    double t_wait = 0, t_copy = 0, t_walk = 0, t_decode = 0, t_intern = 0, t_post = 0;       // SVX_BAM_TIMING=1: seconds per stage, printed at close
    BatchArrays ba[2]; BatchArrays* b = &ba[0];
    Pool* pool = nullptr; Pool* pool_inflate = nullptr;
    bool region_done = false;
    int32_t tid_limit = -2;         // svx_bam_seek: reading stops (like EOF) at the first record whose reference id exceeds this (-2 = none)
    int seq_min_ins = 0;            // > 0: sparse SEQ (coordinate mode): only insertions of at least this length + whole split-read primaries
    struct KeptRange { uint32_t rec; int32_t q0, len; };
    std::vector<std::vector<KeptRange>> t_ranges;          // per decode task
    std::vector<uint32_t> t_rng_cnt; std::vector<uint64_t> t_seq_bytes;
    std::vector<uint32_t> name_id_tmp;
    struct RecRef { const uint8_t* r; const uint8_t* end; const uint8_t* cig; uint32_t n_cig; };
    std::vector<RecRef> refs; std::vector<const char*> t_name, t_sa; std::vector<uint32_t> t_name_len, t_sa_len; std::vector<uint64_t> t_name_hash;
    int64_t total_records = 0;
    // svx_bam_set_gpu_inflate: the GPU inflates sub-batches of blocks from the front of every chunk while the host's cores take blocks from its back
    svx_inflater* gpu = nullptr; size_t gpu_sub = 4096; bool gpu_sub_forced = false;       // SVX_BAM_GPU_SUB: exact sub-batch size (tests)
    int64_t gpu_blocks = 0, cpu_blocks = 0; double gpu_kernel_ms = 0;
    // svx_bam_set_device_decode: inflate, record discovery and decode on the GPU (bamdev.hip); batches come back with device pointers
    svx_devdec* dev = nullptr; int dev_device = -1;
    size_t header_bytes = 0;                  // length of the BAM header in the inflated stream
    size_t dev_fpos = 0; uint64_t dev_skip = 0; bool dev_file_done = false, dev_region_done = false;
    int dev_cur = -1; int64_t dev_first = 0, dev_valid = 0; bool dev_have_carry = false;
    size_t dev_chunk_bytes = (size_t)8192 << 20, dev_chunk_blocks = (size_t)1 << 30;      // test hooks: SVX_BAM_DEV_CHUNK_MB, SVX_BAM_DEV_CHUNK_BLOCKS
    std::string dev_names_blob;
    struct DevLoad { int slot = 0, carry_slot = -1, rc = SVX_OK; std::string err; int64_t n_rec = 0, n_valid = 0; bool file_done = false, empty = false;
                     size_t fpos_start = 0; uint64_t skip = 0; };
    // Chunk slots rotate 0 -> 1 -> 2 -> 0 over the whole life of the reader - ALSO across svx_bam_seek / svx_bam_rewind: the batch handed out last (and the
    // one before it) may still be read by the consumer's stream when the next region's first chunk is loaded (include/svx.h: arrays stay valid until the
    // third next chunk is loaded).  dev_last_slot = the slot that was loaded last.
    int dev_last_slot = -1, dev_handed_slot = -1;      // dev_handed_slot: the slot the last batch was handed out from
    int dev_mode = 0;                         // 0 coordinate-sorted rules, 1 query-name-sorted rules: fixed by the first read after open / rewind / seek
    int dev_grow = 0;                         // a chunk without one complete record is loaded again, into the same slot, with 2^dev_grow times the budget
    size_t dev_region_bytes = 0;              // contig-range reading: budget of the next chunk (small after a seek, x4 per chunk: a range is not read 8 GB beyond its end)
    std::future<DevLoad> dev_future; bool dev_prefetching = false;
};

static void dev_drop_prefetch(svx_bam* h);

// ---- BGZF ------------------------------------------------------------------------------------------------------------------
struct RawBlock { const uint8_t* comp; size_t clen; uint32_t isize; size_t out_at; uint32_t crc; };

static bool read_block(svx_bam* h, RawBlock& b) {
    if (h->fpos >= h->map_len) return false;
    const uint8_t* hd = h->map + h->fpos;
    const size_t left = h->map_len - h->fpos;
    if (left < 18 || hd[0] != 31 || hd[1] != 139 || hd[2] != 8 || !(hd[3] & 4)) throw std::string("not a BGZF block");
    const unsigned xlen = hd[10] | (hd[11] << 8);
    if (left < 12 + (size_t)xlen) throw std::string("truncated BGZF extra field");
    // the BC subfield is first in every BAM writer we know; be tolerant and scan the extra field
    const uint8_t* extra = hd + 12;
    int bsize = -1;
    for (size_t p = 0; p + 4 <= xlen;) {
        const unsigned slen = extra[p + 2] | (extra[p + 3] << 8);
        if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = extra[p + 4] | (extra[p + 5] << 8);
        p += 4 + slen;
    }
    if (bsize < 0) throw std::string("BGZF block without BC subfield");
    if ((size_t)bsize + 1 < 12 + (size_t)xlen + 8 || left < (size_t)bsize + 1) throw std::string("truncated BGZF block");
    const size_t clen = (size_t)bsize + 1 - 12 - xlen - 8;
    b.comp = hd + 12 + xlen; b.clen = clen;
    const uint8_t* tail = b.comp + clen;
    b.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    b.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    h->fpos += (size_t)bsize + 1;
    return true;
}

// one inflate state per worker thread, reset per block (inflateInit2 allocates ~40 KB each time)
struct TlsInflate {
    z_stream zs; bool ok = false;
    ~TlsInflate() { if (ok) inflateEnd(&zs); }
};
static void inflate_block(const RawBlock& b, uint8_t* out) {
    if (b.isize == 0) return;
    static thread_local TlsInflate t;
    if (!t.ok) {
        memset(&t.zs, 0, sizeof t.zs);
        if (inflateInit2(&t.zs, -15) != Z_OK) throw std::string("inflateInit2 failed");
        t.ok = true;
    } else if (inflateReset(&t.zs) != Z_OK) throw std::string("inflateReset failed");
    t.zs.next_in = const_cast<Bytef*>(b.comp); t.zs.avail_in = (uInt)b.clen;
    t.zs.next_out = out; t.zs.avail_out = b.isize;
    const int rc = inflate(&t.zs, Z_FINISH);
    if (rc != Z_STREAM_END || t.zs.avail_out != 0) throw std::string("BGZF inflate failed");
    // SVX_BAM_VERIFY_CRC=1: the CRC32 of the block trailer, as htslib checks it (off by default: it costs about as much as a third of the inflate;
    // blocks the GPU inflates are checked for their length and for a sound DEFLATE stream only)
    static const bool verify = []() { const char* e = getenv("SVX_BAM_VERIFY_CRC"); return e && e[0] == '1'; }();
    if (verify && (uint32_t)crc32(crc32(0L, Z_NULL, 0), out, b.isize) != b.crc) throw std::string("BGZF block fails its CRC32");
}

// Inflate the next chunk of BGZF blocks (<= 1024 blocks / 48 MB) into h->next: n_threads workers, runs on a background thread while the
// caller decodes the previous chunk.  Only this function touches the FILE after open.
static void inflate_next_chunk(svx_bam* h) {
    const size_t WIN_HEAD = h->win_head;
    try {
        std::vector<RawBlock> blocks;
        size_t total = 0;
        h->next_eof = false;
        const size_t max_bytes = h->chunk_bytes;
        while (blocks.size() < h->chunk_blocks && total < max_bytes) {
            RawBlock b;
            if (!read_block(h, b)) { h->next_eof = true; break; }
            b.out_at = total; total += b.isize;
            blocks.push_back(b);
        }
        h->next.resize_uninit(WIN_HEAD + total);
        h->next_len = total;
        if (h->gpu && !blocks.empty()) {
            // Both ends against the middle: a feeder thread hands sub-batches from the FRONT of the chunk to the GPU (payloads packed into pinned staging,
            // three sub-batches in flight: H2D, inflate and the copy back overlap), the worker threads inflate runs of 8 blocks from the BACK with zlib;
            // whoever is faster takes more.  (The window is pageable memory of the reader: the inflated bytes reach it through the inflater's page-locked buffer.)
            uint8_t* out_base = h->next.data() + WIN_HEAD;
            std::mutex m;
            size_t lo = 0, hi = blocks.size();
            auto take = [&](bool front, size_t want, size_t& a, size_t& b) -> bool {
                std::lock_guard<std::mutex> g(m);
                if (lo >= hi) return false;
                if (front) { a = lo; b = std::min(hi, lo + want); lo = b; }
                else { b = hi; a = hi > lo + want ? hi - want : lo; hi = a; }
                return true;
            };
            std::string gpu_err;
            int64_t n_gpu = 0; double ms_gpu = 0;
            std::thread feeder([&]() {
                bool used[3] = {false, false, false};
                auto wait_slot = [&](int sl) { float ms = 0; if (svx_inflater_wait(h->gpu, sl, &ms) != SVX_OK) throw std::string(svx_last_error()); ms_gpu += ms; used[sl] = false; };
                try {
                    std::vector<uint64_t> in_off, out_at; std::vector<uint32_t> clen, isize;
                    size_t a, b; int slot = 0;
                    // sub-batch: as large as possible for the GPU's sake (thousands of wavefronts), at most half of the chunk so that the cores get their share
                    const size_t sub = h->gpu_sub_forced ? h->gpu_sub : std::min(h->gpu_sub, std::max<size_t>(2048, blocks.size() / 2));
                    while (take(true, sub, a, b)) {
                        if (used[slot]) wait_slot(slot);
                        const size_t n = b - a;
                        in_off.resize(n); out_at.resize(n); clen.resize(n); isize.resize(n);
                        // the blocks of a sub-batch are one contiguous slice of the file: it goes to the pinned staging buffer as it is (headers and
                        // trailers included), in four concurrent copies
                        const uint8_t* f0 = blocks[a].comp;
                        const uint64_t staged = (uint64_t)(blocks[b - 1].comp + blocks[b - 1].clen - f0);
                        const size_t base = blocks[a].out_at;
                        for (size_t i = 0; i < n; i++) {
                            const RawBlock& rb = blocks[a + i];
                            in_off[i] = (uint64_t)(rb.comp - f0);
                            clen[i] = (uint32_t)rb.clen; isize[i] = rb.isize; out_at[i] = rb.out_at - base;
                        }
                        uint8_t* stage = (uint8_t*)svx_inflater_staging(h->gpu, slot, staged + 8);
                        if (!stage) throw std::string("no pinned staging memory");
                        {
                            const int parts = staged > ((uint64_t)8 << 20) ? 4 : 1;
                            std::vector<std::thread> cp;
                            for (int q = 1; q < parts; q++) {
                                const uint64_t lo_b = staged * (uint64_t)q / parts, hi_b = staged * (uint64_t)(q + 1) / parts;
                                cp.emplace_back([=]() { memcpy(stage + lo_b, f0 + lo_b, (size_t)(hi_b - lo_b)); });
                            }
                            memcpy(stage, f0, (size_t)(staged / parts));
                            for (auto& t : cp) t.join();
                        }
                        const uint64_t out_bytes = blocks[b - 1].out_at + blocks[b - 1].isize - base;
                        if (svx_inflater_enqueue(h->gpu, slot, (int64_t)n, in_off.data(), clen.data(), isize.data(), out_at.data(), staged, out_base + base, out_bytes, 0) != SVX_OK)
                            throw std::string(svx_last_error());
                        used[slot] = true; n_gpu += (int64_t)n;
                        slot = (slot + 1) % 3;
                    }
                    for (int sl = 0; sl < 3; sl++) if (used[sl]) wait_slot(sl);
                } catch (const std::string& e) {
                    gpu_err = e;
                    for (int sl = 0; sl < 3; sl++) if (used[sl]) (void)svx_inflater_wait(h->gpu, sl, nullptr);
                } catch (const std::exception& e) {
                    gpu_err = std::string("GPU inflate feeder: ") + e.what();
                    for (int sl = 0; sl < 3; sl++) if (used[sl]) (void)svx_inflater_wait(h->gpu, sl, nullptr);
                }
            });
            const int n_workers = std::max(1, h->n_threads);
            std::vector<std::string> errs((size_t)n_workers);
            std::vector<int64_t> done((size_t)n_workers, 0);
            h->pool_inflate->run(n_workers, [&](int t) {
                try { size_t a, b; while (take(false, 8, a, b)) { for (size_t i = a; i < b; i++) inflate_block(blocks[i], out_base + blocks[i].out_at); done[(size_t)t] += (int64_t)(b - a); } }
                catch (const std::string& e) { errs[(size_t)t] = e; }
                catch (const std::exception& e) { errs[(size_t)t] = std::string("inflate worker: ") + e.what(); }
            });
            feeder.join();
            h->gpu_blocks += n_gpu; h->gpu_kernel_ms += ms_gpu;
            for (auto d : done) h->cpu_blocks += d;
            if (!gpu_err.empty()) throw gpu_err;
            for (auto& e : errs) if (!e.empty()) throw e;
            return;
        }
        // blocks in runs of 8 per task: dynamic scheduling over the pool evens out the cost differences between blocks
        const int n_tasks = (int)((blocks.size() + 7) / 8);
        std::vector<std::string> errs((size_t)std::max(1, n_tasks));
        h->pool_inflate->run(n_tasks, [&](int t) {
            try { for (size_t i = (size_t)t * 8; i < blocks.size() && i < (size_t)(t + 1) * 8; i++) inflate_block(blocks[i], h->next.data() + WIN_HEAD + blocks[i].out_at); }
            catch (const std::string& e) { errs[(size_t)t] = e; }
            catch (const std::exception& e) { errs[(size_t)t] = std::string("inflate worker: ") + e.what(); }
        });
        for (auto& e : errs) if (!e.empty()) throw e;
    } catch (const std::string& e) { h->prefetch_err = e; }
      catch (const std::exception& e) { h->prefetch_err = std::string("inflate: ") + e.what(); }
}

static void start_prefetch(svx_bam* h) {
    if (h->prefetch_active || h->file_eof) return;
    h->prefetch = std::async(std::launch::async, inflate_next_chunk, h);
    h->prefetch_active = true;
}

// make at least `need` bytes available from h->pos (false at clean EOF with nothing left)
static bool ensure(svx_bam* h, size_t need) {
    while (h->buf.size() - h->pos < need) {
        if (h->file_eof) return false;
        start_prefetch(h);
        double t0 = now_s();
        h->prefetch.get();
        h->prefetch_active = false;
        h->t_wait += now_s() - t0; t0 = now_s();
        if (!h->prefetch_err.empty()) { const std::string e = h->prefetch_err; h->prefetch_err.clear(); h->file_eof = true; throw e; }
        // the unconsumed tail of the current window (a partial record) moves in front of the new chunk; the old buffer becomes the
        // next inflate target
        const size_t WIN_HEAD = h->win_head;
        const size_t keep = h->buf.size() - h->pos;
        if (keep <= WIN_HEAD) {
            if (keep) memcpy(h->next.data() + WIN_HEAD - keep, h->buf.data() + h->pos, keep);
            std::swap(h->buf.p, h->next.p); std::swap(h->buf.cap, h->next.cap);
            h->next.n = 0;                                   // the old window's content is dead: nothing to preserve when it grows
            h->buf.n = WIN_HEAD + h->next_len;
            h->pos = WIN_HEAD - keep;
        } else {
            // a record longer than the headroom straddles the chunks: fall back to appending
            if (h->pos) { memmove(h->buf.data(), h->buf.data() + h->pos, keep); h->buf.n = keep; h->pos = 0; }
            h->buf.append(h->next.data() + WIN_HEAD, h->next_len);
        }
        if (h->next_eof) h->file_eof = true;
        const bool got = h->next_len > 0;
        h->t_copy += now_s() - t0;
        start_prefetch(h);
        if (!got && h->file_eof) break;
    }
    return h->buf.size() - h->pos >= need;
}

static inline uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// CPUs this process may actually use: the visible cores capped by the cgroup CPU quota (a container that shows 256 cores but is granted
// 16 CPUs' worth of time gets throttled for the rest of every scheduler period once 100+ busy threads have burnt the quota)
static int granted_cpus() {                                 // CPUs the cgroup quota grants (the visible cores when there is none)
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long period = 0; char first[32] = {0};
        if (fscanf(f, "%31s %lld", first, &period) == 2 && strcmp(first, "max") != 0 && period > 0) { const long long q = (atoll(first) + period - 1) / period; if (q >= 1 && (unsigned)q < n) n = (unsigned)q; }
        fclose(f);
    }
    return (int)n;
}
static int default_threads() {
    unsigned n = std::max(1u, std::min(128u, std::thread::hardware_concurrency()));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = -1, period = 0;
        char first[32] = {0};
        if (fscanf(f, "%31s %lld", first, &period) == 2 && strcmp(first, "max") != 0) quota = atoll(first);
        fclose(f);
        // a CPU quota (container): 4 threads per granted CPU - the quota is enforced per 100 ms period, and a pool as wide as the quota leaves
        // granted time unused whenever a worker waits (measured on a 16-CPU grant: 16 threads 0.36, 64 threads 0.47 M records/s)
        if (quota > 0 && period > 0) { const unsigned q = 4 * (unsigned)((quota + period - 1) / period); if (q >= 1 && q < n) n = q; }
    }
    return (int)n;
}

// BGZF inflate shared between the GPU (svx_inflater, device >= 0) and the host's cores; device < 0 switches it off again.  Larger chunks then:
// the GPU wants thousands of blocks in flight.
static void drop_gpu(svx_bam* h) {
    if (!h->gpu) return;
    svx_inflater_destroy(h->gpu);
    h->gpu = nullptr;
}

extern "C" int svx_bam_set_gpu_inflate(svx_bam* h, int device) {
    if (!h) return bam_fail(SVX_E_ARG, "null handle");
    if (h->prefetch_active) h->prefetch.wait();          // the chunk being inflated stays what it is; the next one sees the new setting
    drop_gpu(h);
    if (device < 0) return SVX_OK;
    const int rc = svx_inflater_create(device, &h->gpu);
    if (rc != SVX_OK) { h->gpu = nullptr; return rc; }
    if (!getenv("SVX_BAM_CHUNK_BLOCKS")) {                   // (the test hook keeps its small chunks)
        h->chunk_bytes = std::max<size_t>(h->chunk_bytes, (size_t)3072 << 20);
        h->chunk_blocks = std::max<size_t>(h->chunk_blocks, 65536);
    }
    h->gpu_sub = 32768;
    h->gpu_sub_forced = false;
    { const char* e = getenv("SVX_BAM_GPU_SUB"); if (e && atoll(e) > 0) { h->gpu_sub = (size_t)atoll(e); h->gpu_sub_forced = true; } }
    { const char* e = getenv("SVX_BAM_GPU_CHUNK_MB"); if (e && atoll(e) > 0) { h->chunk_bytes = (size_t)atoll(e) << 20; h->chunk_blocks = h->chunk_bytes / 30000; } }
    return SVX_OK;
}
extern "C" int svx_bam_gpu_inflate_stats(svx_bam* h, int64_t* gpu_blocks, int64_t* cpu_blocks, double* gpu_kernel_ms) {
    if (!h) return bam_fail(SVX_E_ARG, "null handle");
    if (h->dev) {                                            // device decode: every block is inflated by the GPU
        DevDecStats ds; devdec_stats(h->dev, &ds);
        if (gpu_blocks) *gpu_blocks = ds.gpu_blocks;
        if (cpu_blocks) *cpu_blocks = ds.cpu_blocks;
        if (gpu_kernel_ms) *gpu_kernel_ms = ds.inflate_kernel_ms;
        return SVX_OK;
    }
    if (gpu_blocks) *gpu_blocks = h->gpu_blocks;
    if (cpu_blocks) *cpu_blocks = h->cpu_blocks;
    if (gpu_kernel_ms) *gpu_kernel_ms = h->gpu_kernel_ms;
    return SVX_OK;
}

extern "C" int svx_bam_open(const char* path, int n_threads, svx_bam** out) {
    svx_bam* h = new svx_bam();
    h->path = path;
    h->n_threads = n_threads > 0 ? n_threads : default_threads();
    // a chunk must keep every worker busy for a while: ~4 MB of inflated data per thread
    h->chunk_bytes = std::max<size_t>((size_t)48 << 20, (size_t)h->n_threads * ((size_t)4 << 20));
    h->chunk_blocks = std::max<size_t>(1024, h->chunk_bytes / 48000);
    { const char* e = getenv("SVX_BAM_WIN_HEAD"); if (e && atoll(e) >= 0) h->win_head = (size_t)atoll(e); }
    { const char* e = getenv("SVX_BAM_CHUNK_BLOCKS"); if (e && atoll(e) > 0) h->chunk_blocks = (size_t)atoll(e); }
    h->pool = new Pool(h->n_threads - 1);                 // record decode (the calling thread takes part)
    h->pool_inflate = new Pool(h->n_threads - 1);         // BGZF inflate of the NEXT chunk, concurrently with the decode of this one
    h->fd = open(path, O_RDONLY);
    struct stat sb;
    if (h->fd < 0 || fstat(h->fd, &sb) != 0) { if (h->fd >= 0) close(h->fd); delete h; return bam_fail(SVX_E_ARG, std::string("cannot open ") + path); }
    h->map_len = (size_t)sb.st_size;
    if (h->map_len) {
        void* m = mmap(nullptr, h->map_len, PROT_READ, MAP_PRIVATE, h->fd, 0);
        if (m == MAP_FAILED) { close(h->fd); delete h; return bam_fail(SVX_E_ARG, std::string("cannot map ") + path); }
        h->map = (const uint8_t*)m;
        (void)madvise(m, h->map_len, MADV_SEQUENTIAL);
    }
    try {
        if (!ensure(h, 12) || memcmp(h->buf.data() + h->pos, "BAM\1", 4) != 0) throw std::string("not a BAM file");
        const uint32_t l_text = rd32(h->buf.data() + h->pos + 4);
        if (!ensure(h, 12 + (size_t)l_text)) throw std::string("truncated BAM header");
        std::string text((const char*)h->buf.data() + h->pos + 8, l_text);
        h->pos += 8 + (size_t)l_text;
        // @HD SO:
        const size_t hd = text.find("@HD");
        if (hd != std::string::npos) {
            const size_t eol = text.find('\n', hd);
            const std::string line = text.substr(hd, eol == std::string::npos ? std::string::npos : eol - hd);
            const size_t so = line.find("\tSO:");
            if (so != std::string::npos) { size_t e = line.find('\t', so + 4); h->sort_order = line.substr(so + 4, e == std::string::npos ? std::string::npos : e - so - 4); }
        }
        const uint32_t n_ref = rd32(h->buf.data() + h->pos); h->pos += 4;
        for (uint32_t i = 0; i < n_ref; i++) {
            if (!ensure(h, 4)) throw std::string("truncated BAM reference list");
            const uint32_t l_name = rd32(h->buf.data() + h->pos); h->pos += 4;
            if (!ensure(h, l_name + 4)) throw std::string("truncated BAM reference list");
            std::string nm((const char*)h->buf.data() + h->pos, l_name ? l_name - 1 : 0); h->pos += l_name;
            h->ref_len.push_back((int32_t)rd32(h->buf.data() + h->pos)); h->pos += 4;
            h->tid_of.emplace(nm, (int32_t)i);
            h->names_blob += nm; h->names_blob.push_back('\0');
            h->ref_names.push_back(std::move(nm));
        }
        // rank of each contig NAME in Python str order (bytewise for ASCII names)
        std::vector<int32_t> idx(n_ref);
        for (uint32_t i = 0; i < n_ref; i++) idx[i] = (int32_t)i;
        std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return h->ref_names[(size_t)a] < h->ref_names[(size_t)b]; });
        h->contig_rank.assign(n_ref ? n_ref : 1, 0);
        for (uint32_t r = 0; r < n_ref; r++) h->contig_rank[(size_t)idx[r]] = (int32_t)r;
        h->header_bytes = h->pos - h->win_head;                      // (the header sits in the first window, which starts at win_head)
    } catch (const std::string& e) {
        if (h->prefetch_active) h->prefetch.wait();
        if (h->map) munmap((void*)h->map, h->map_len);
        close(h->fd); delete h->pool; delete h->pool_inflate; delete h;
        return bam_fail(SVX_E_ARG, e);
    }
    *out = h;
    return SVX_OK;
}

extern "C" void svx_bam_close(svx_bam* h) {
    if (!h) return;
    if (h->prefetch_active) { h->prefetch.wait(); h->prefetch_active = false; }
    if (getenv("SVX_BAM_TIMING"))
        fprintf(stderr, "bamio %d threads: wait for inflate %.3f s, window copy %.3f, record walk %.3f, decode %.3f, names + SA %.3f\n", h->n_threads,
                h->t_wait, h->t_copy, h->t_walk, h->t_decode, h->t_intern);
    if (h->gpu) {
        if (getenv("SVX_BAM_TIMING")) fprintf(stderr, "bamio inflate: %lld blocks on the GPU (kernels %.1f ms), %lld on the host\n", (long long)h->gpu_blocks, h->gpu_kernel_ms, (long long)h->cpu_blocks);
        drop_gpu(h);                                     // before the buffers it page-locked are freed
    }
    if (h->dev) {
        dev_drop_prefetch(h);
        if (getenv("SVX_BAM_TIMING")) {
            DevDecStats ds; devdec_stats(h->dev, &ds);
            fprintf(stderr, "bamio device decode: %lld blocks (%lld GPU, %lld host), %.1f MB inflated, %lld records; staging + enqueue %.3f s, waiting for the inflate %.3f (kernels %.1f ms), "
                            "record discovery %.3f, decode %.3f, names %.3f; %lld serial fallbacks\n", (long long)ds.blocks, (long long)ds.gpu_blocks, (long long)ds.cpu_blocks, ds.bytes / 1e6, (long long)ds.records, ds.t_stage,
                    ds.t_inflate_wait, ds.inflate_kernel_ms, ds.t_discover, ds.t_decode, ds.t_names, (long long)ds.fallbacks);
        }
        devdec_destroy(h->dev); h->dev = nullptr;
    }
    if (h->map) munmap((void*)h->map, h->map_len);
    if (h->fd >= 0) close(h->fd);
    delete h->pool; delete h->pool_inflate;
    delete h;
}

// Sparse SEQ for coordinate-sorted input: COLLECT reads a record's bases only for the insertions it reports (CIGAR I of at least
// min_sv_size, src/svim/SVIM_intra.py:24-27,41-47) and - for a primary with an SA tag - for the read gap between two segments
// (src/svim/SVIM_inter.py:85,91).  With min_ins_len > 0 the reader keeps exactly those ranges (whole SEQ for records carrying an SA
// tag) and describes them in svx_batch.seq_rng_*; about 2 % of the bases of an ONT file survive, which is what then crosses PCIe.
// 0 (default): every record's full SEQ, seq_rng_off = NULL.  Query-name mode always keeps everything.
extern "C" int svx_bam_set_seq_filter(svx_bam* h, int min_ins_len) {
    if (!h) return bam_fail(SVX_E_ARG, "null reader");
    h->seq_min_ins = min_ins_len > 0 ? min_ins_len : 0;
    return SVX_OK;
}

// Contig-range reading for contig-sharded ranks: continue at virtual offset `voff` (from the .bai: the first record of a contig) and
// treat the first record whose reference id is above `last_tid` (or unplaced) as the end of the file.  last_tid = -2 lifts the limit.
extern "C" int svx_bam_seek(svx_bam* h, uint64_t voff, int32_t last_tid) {
    if (!h) return bam_fail(SVX_E_ARG, "null reader");
    if (h->prefetch_active) { h->prefetch.wait(); h->prefetch_active = false; }
    h->prefetch_err.clear();
    const size_t coff = (size_t)(voff >> 16), uoff = (size_t)(voff & 0xffff);
    if (coff > h->map_len) return bam_fail(SVX_E_ARG, "virtual offset beyond the end of the file");
    h->fpos = coff; h->file_eof = false; h->buf.clear(); h->pos = 0; h->next_len = 0; h->next_eof = false;
    h->tid_limit = last_tid; h->region_done = false;
    if (h->dev) {                                          // device decode: the next chunk starts at that block, `uoff` bytes into its data
        dev_drop_prefetch(h);
        h->dev_fpos = coff; h->dev_skip = uoff; h->dev_file_done = false; h->dev_region_done = false; h->dev_cur = -1; h->dev_first = h->dev_valid = 0; h->dev_have_carry = false;
        h->dev_grow = 0; h->dev_region_bytes = last_tid != -2 ? (size_t)256 << 20 : 0;
        return SVX_OK;
    }
    try {
        if (uoff && !ensure(h, uoff)) throw std::string("virtual offset beyond its block");
        h->pos += uoff;
    } catch (const std::string& e) { return bam_fail(SVX_E_ARG, e); }
    return SVX_OK;
}

// Back to the first alignment record, keeping every buffer, thread and interned read name: a second pass over the same file (the
// steady state of a long file: no first-touch allocation anywhere).
extern "C" int svx_bam_rewind(svx_bam* h) {
    if (!h) return bam_fail(SVX_E_ARG, "null reader");
    if (h->prefetch_active) { h->prefetch.wait(); h->prefetch_active = false; }
    h->prefetch_err.clear();
    if (getenv("SVX_BAM_TIMING")) {                      // per pass: the stages of the pass that just ended
        fprintf(stderr, "bamio pass: wait for inflate %.3f s, window copy %.3f, record walk %.3f, decode %.3f, names + SA %.3f\n", h->t_wait, h->t_copy, h->t_walk, h->t_decode, h->t_intern);
        h->t_wait = h->t_copy = h->t_walk = h->t_decode = h->t_intern = 0;
    }
    if (h->dev) {                                          // device decode: nothing to inflate here - the first chunk is loaded by the first read
        dev_drop_prefetch(h);
        h->dev_fpos = 0; h->dev_skip = h->header_bytes; h->dev_file_done = false; h->dev_region_done = false; h->dev_cur = -1; h->dev_first = h->dev_valid = 0; h->dev_have_carry = false;
        h->dev_grow = 0; h->dev_region_bytes = 0;
        h->total_records = 0; h->tid_limit = -2; h->region_done = false;
        return SVX_OK;
    }
    h->fpos = 0; h->file_eof = false; h->buf.clear(); h->pos = 0; h->next_len = 0; h->next_eof = false;
    try {
        if (!ensure(h, 12)) throw std::string("not a BAM file");
        const uint32_t l_text = rd32(h->buf.data() + h->pos + 4);
        if (!ensure(h, 12 + (size_t)l_text)) throw std::string("truncated BAM header");
        h->pos += 8 + (size_t)l_text;
        const uint32_t n_ref = rd32(h->buf.data() + h->pos); h->pos += 4;
        for (uint32_t i = 0; i < n_ref; i++) {
            if (!ensure(h, 4)) throw std::string("truncated BAM reference list");
            const uint32_t l_name = rd32(h->buf.data() + h->pos); h->pos += 4;
            if (!ensure(h, l_name + 4)) throw std::string("truncated BAM reference list");
            h->pos += l_name + 4;
        }
    } catch (const std::string& e) { return bam_fail(SVX_E_ARG, e); }
    h->total_records = 0; h->tid_limit = -2; h->region_done = false;
    return SVX_OK;
}

extern "C" int svx_bam_header(svx_bam* h, int32_t* n_ref, const char** names_blob, const int32_t** lengths, const char** sort_order) {
    *n_ref = (int32_t)h->ref_names.size(); *names_blob = h->names_blob.c_str(); *lengths = h->ref_len.data(); *sort_order = h->sort_order.c_str();
    return SVX_OK;
}

static bool parse_int(const char* s, size_t n, long long& v) {
    if (n == 0) return false;
    size_t i = 0; bool neg = false;
    if (s[0] == '-' || s[0] == '+') { neg = s[0] == '-'; i = 1; if (n == 1) return false; }
    long long x = 0;
    for (; i < n; i++) { if (s[i] < '0' || s[i] > '9') return false; x = x * 10 + (s[i] - '0'); if (x > (1ll << 40)) return false; }
    v = neg ? -x : x;
    return true;
}

// SA tag -> segment rows (src/svim/SVIM_COLLECT.py:55-85): 6-field check, pos-1, strand, mapq overflow -> 0
static int append_sa(svx_bam* h, const char* sa, size_t len, int32_t primary_lseq) {
    size_t p = 0;
    while (p < len) {
        size_t e = p; while (e < len && sa[e] != ';') e++;
        if (e > p) {
            size_t fs[7]; int nf = 0; fs[0] = p;
            for (size_t i = p; i < e; i++) if (sa[i] == ',') { if (nf < 6) fs[++nf] = i + 1; else { nf++; } }
            if (nf == 5) {
                fs[6] = e + 1;
                auto fld = [&](int k, const char*& s, size_t& n) { s = sa + fs[k]; n = fs[k + 1] - 1 - fs[k]; };
                const char* s; size_t n; long long pos, mq, nm;
                fld(0, s, n); const std::string rname(s, n);
                fld(1, s, n); if (!parse_int(s, n, pos)) return bam_fail(SVX_E_ARG, "malformed SA tag (position)");
                fld(2, s, n); const bool rev = !(n == 1 && s[0] == '+');
                const char* cs; size_t cn; fld(3, cs, cn);
                fld(4, s, n); if (!parse_int(s, n, mq)) return bam_fail(SVX_E_ARG, "malformed SA tag (mapq)");
                fld(5, s, n); if (!parse_int(s, n, nm)) return bam_fail(SVX_E_ARG, "malformed SA tag (NM)");
                if (mq < 0 || mq > 255) mq = 0;
                auto it = h->tid_of.find(rname);
                h->b->seg_tid.push_back(it == h->tid_of.end() ? -1 : it->second);
                h->b->seg_pos.push_back((int32_t)(pos - 1)); h->b->seg_rev.push_back(rev ? 1 : 0); h->b->seg_mapq.push_back((uint8_t)mq);
                h->b->seg_lseq.push_back(primary_lseq);
                long long num = 0; bool have = false;
                if (!(cn == 1 && cs[0] == '*')) {
                    for (size_t i = 0; i < cn; i++) {
                        const char ch = cs[i];
                        if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); have = true; }
                        else {
                            const char* ops = "MIDNSHP=XB"; const char* q = strchr(ops, ch);
                            if (!q || !have) return bam_fail(SVX_E_ARG, "malformed CIGAR in SA tag");
                            h->b->seg_cigar.push_back((uint32_t)(num << 4) | (uint32_t)(q - ops)); num = 0; have = false;
                        }
                    }
                    if (have) return bam_fail(SVX_E_ARG, "malformed CIGAR in SA tag");
                }
                h->b->seg_cigar_off.push_back(h->b->seg_cigar.size());
            } else {
                fprintf(stderr, "WARNING: SA tag does not consist of 6 fields. This could be a sign of invalid characters "
                                "(e.g. commas or semicolons) in a chromosome name of the reference genome.\n");
            }
        }
        p = e + 1;
    }
    return SVX_OK;
}

// does the record's CIGAR field stand in for a real CIGAR kept in the CG tag?  (htslib sam.c bam_tag2cigar)
static inline bool cg_placeholder(const uint8_t* rec, const uint8_t* cig, uint32_t n_cig, uint32_t l_seq) {
    return n_cig >= 1 && (int32_t)rd32(rec) >= 0 && (int32_t)rd32(rec + 4) >= 0 && (rd32(cig) & 15) == 4 && (rd32(cig) >> 4) == l_seq;
}

// aux fields of one record: SA (Z) and CG (B,I).  CG as htslib's bam_aux_get finds it: the FIRST field of that name, whatever its type - a first CG of
// another type hides a later array (bam_tag2cigar then leaves the record alone)
static void scan_aux(const uint8_t* q, const uint8_t* end, const char*& sa, size_t& sa_n, const uint8_t*& cg, uint32_t& cg_n) {
    sa = nullptr; sa_n = 0; cg = nullptr; cg_n = 0;
    bool cg_seen = false, sa_seen = false;      // bam_aux_get (pysam's get_tag, bam_tag2cigar): the FIRST field of a name counts, whatever its type
    while (q + 3 <= end) {
        const char t0 = (char)q[0], t1 = (char)q[1], ty = (char)q[2]; q += 3;
        size_t sz = 0;
        const size_t left = (size_t)(end - q);
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { const uint8_t* z = (const uint8_t*)memchr(q, 0, (size_t)(end - q)); if (!z) throw std::string("unterminated aux string");
                if (t0 == 'S' && t1 == 'A' && ty == 'Z' && !sa_seen) { sa = (const char*)q; sa_n = (size_t)(z - q); } sz = (size_t)(z - q) + 1; break; }
            case 'B': { if (left < 5) throw std::string("truncated BAM aux array");
                const char sub = (char)q[0]; const uint32_t cnt = rd32(q + 1); const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                sz = 5 + es * (size_t)cnt;
                if (sz > left) throw std::string("BAM aux array runs past the end of its record");
                if (t0 == 'C' && t1 == 'G' && !cg_seen && (sub == 'I' || sub == 'i') && cnt > 0) { cg = q + 5; cg_n = cnt; } break; }
            default: throw std::string("unknown BAM aux type");
        }
        if (t0 == 'C' && t1 == 'G') cg_seen = true;
        if (t0 == 'S' && t1 == 'A') sa_seen = true;
        if (sz > left) throw std::string("BAM aux field runs past the end of its record");
        q += sz;
    }
}

// Bulk decode: every complete record in the window (at most max_count) -> batch arrays.  (1) a serial walk over the block_size fields
// finds the records and sizes the CIGAR array; (2) the pool decodes the records in runs of 256 (numeric fields, CIGAR words, SA / name
// pointers) and plans which SEQ ranges to keep; (3) prefix sums over the plan; (4) the pool copies the kept bases; (5) serial: read names
// are interned in record order and the SA strings copied.  Returns the number of records decoded (0: no complete record in the window).
static int64_t decode_run(svx_bam* h, int64_t max_count, bool sparse) {
    auto& refs = h->refs;
    refs.clear();
    double t0 = now_s();
    BatchArrays& B = *h->b;
    const uint8_t* buf = h->buf.data();
    size_t p = h->pos;
    const size_t end = h->buf.size();
    uint64_t cig_total = B.cigar.size();
    while ((int64_t)refs.size() < max_count && p + 4 <= end) {
        const uint32_t bs = rd32(buf + p);
        if (p + 4 + (size_t)bs > end) break;
        svx_bam::RecRef rr;
        rr.r = buf + p + 4; rr.end = rr.r + bs;
        if (bs < 32) throw std::string("corrupt BAM record");
        const unsigned l_name = rr.r[8];
        rr.n_cig = rd16(rr.r + 12);
        const uint32_t l_seq = rd32(rr.r + 16);
        rr.cig = rr.r + 32 + l_name;
        if (rr.cig + 4 * (size_t)rr.n_cig + (l_seq + 1) / 2 + l_seq > rr.end) throw std::string("corrupt BAM record");
        if (h->tid_limit != -2) {                          // contig-range reading: the range ends where the next contig (or the unplaced tail) begins
            const int32_t t = (int32_t)rd32(rr.r);
            if (t < 0 || t > h->tid_limit) { h->region_done = true; break; }
        }
        // long CIGARs (> 65535 ops) live in CG:B,I behind a placeholder whose first operation clips the whole read - htslib's bam_tag2cigar rule: mapped
        // record (tid, pos >= 0), first operation <l_seq>S, CG of type B,I / B,i with at least one word; the rest of the placeholder does not matter
        if (cg_placeholder(rr.r, rr.cig, rr.n_cig, l_seq)) {
            const char* sa; size_t sa_n; const uint8_t* cg; uint32_t cg_n;
            scan_aux(rr.cig + 4 * (size_t)rr.n_cig + (l_seq + 1) / 2 + l_seq, rr.end, sa, sa_n, cg, cg_n);
            // (bam_tag2cigar: only a CG array at least as long as the placeholder replaces it - "don't move if the real CIGAR length is shorter than the fake")
            if (cg && cg_n >= rr.n_cig && cg_n < (1u << 29)) { if (cg + 4 * (size_t)cg_n > rr.end) throw std::string("corrupt CG tag"); rr.cig = cg; rr.n_cig = cg_n; }
        }
        cig_total += rr.n_cig;
        B.cigar_off.push_back(cig_total);
        refs.push_back(rr);
        p += 4 + (size_t)bs;
    }
    const size_t n = refs.size();
    if (n == 0) return 0;
    h->t_walk += now_s() - t0; t0 = now_s();
    const size_t base = B.flag.size();
    B.flag.resize(base + n); B.tid.resize(base + n); B.bpos.resize(base + n); B.mapq.resize(base + n); B.lseq.resize(base + n);
    B.read_id.resize(base + n); B.sa_at.resize(base + n); B.sa_len.resize(base + n);
    h->t_name.resize(n); h->t_name_len.resize(n); h->t_sa.resize(n); h->t_sa_len.resize(n); h->t_name_hash.resize(n);
    B.cigar.resize_uninit(cig_total);
    const size_t RUN = 256;
    const int n_tasks = (int)((n + RUN - 1) / RUN);
    h->t_ranges.resize((size_t)n_tasks);
    h->t_rng_cnt.assign(n, 0); h->t_seq_bytes.assign(n, 0);
    std::vector<std::string> errs((size_t)n_tasks);
    const int min_ins = h->seq_min_ins;
    h->pool->run(n_tasks, [&](int t) {
        try {
            auto& kept = h->t_ranges[(size_t)t];
            kept.clear();
            for (size_t i = (size_t)t * RUN; i < n && i < (size_t)(t + 1) * RUN; i++) {
                const svx_bam::RecRef& rr = refs[i];
                const uint8_t* r = rr.r;
                const size_t k = base + i;
                const unsigned l_name = r[8];
                const uint32_t l_seq = rd32(r + 16);
                B.tid[k] = (int32_t)rd32(r); B.bpos[k] = (int32_t)rd32(r + 4); B.mapq[k] = r[9];
                B.flag[k] = (uint16_t)(rd16(r + 14) & 0x0fff); B.lseq[k] = (int32_t)l_seq;
                h->t_name[i] = (const char*)r + 32; h->t_name_len[i] = l_name ? l_name - 1 : 0;
                h->t_name_hash[i] = name_hash(h->t_name[i], h->t_name_len[i]);
                const uint8_t* q = r + 32 + l_name + 4 * (size_t)rd16(r + 12);       // the record's own CIGAR field, CG or not
                const char* sa; size_t sa_n; const uint8_t* cg; uint32_t cg_n;
                scan_aux(q + (l_seq + 1) / 2 + l_seq, rr.end, sa, sa_n, cg, cg_n);
                h->t_sa[i] = sa; h->t_sa_len[i] = (uint32_t)sa_n;
                uint32_t* cw = B.cigar.data() + B.cigar_off[k];
                if (!sparse || sa_n || l_seq == 0) {
                    for (uint32_t c = 0; c < rr.n_cig; c++) cw[c] = rd32(rr.cig + 4 * (size_t)c);
                    if (l_seq) { kept.push_back({(uint32_t)i, 0, (int32_t)l_seq}); h->t_rng_cnt[i] = 1; h->t_seq_bytes[i] = (l_seq + 1) / 2; }
                } else {
                    // walk the query offset along the CIGAR (M, I, S, =, X consume stored bases) and keep what a reported insertion reads
                    uint32_t qpos = 0, cnt = 0; uint64_t bytes = 0;
                    for (uint32_t c = 0; c < rr.n_cig; c++) {
                        const uint32_t w = rd32(rr.cig + 4 * (size_t)c);
                        cw[c] = w;
                        const uint32_t op = w & 15u, len = w >> 4;
                        if (op == 1 && (int)len >= min_ins) {
                            const uint32_t q0 = qpos & ~1u;                              // whole bytes: ranges start at an even base
                            uint32_t q1 = qpos + len; if (q1 > l_seq) q1 = l_seq;
                            if (q1 > q0) { kept.push_back({(uint32_t)i, (int32_t)q0, (int32_t)(q1 - q0)}); cnt++; bytes += (q1 - q0 + 1) / 2; }
                        }
                        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qpos += len;
                    }
                    h->t_rng_cnt[i] = cnt; h->t_seq_bytes[i] = bytes;
                }
            }
        } catch (const std::string& e) { errs[(size_t)t] = e; }
          catch (const std::exception& e) { errs[(size_t)t] = std::string("decode worker: ") + e.what(); }
    });
    for (auto& e : errs) if (!e.empty()) throw e;
    // (3) offsets of the kept ranges and their bytes
    uint64_t seq_total = B.seq.size();
    uint32_t rng_total = (uint32_t)B.rng_q0.size();
    for (size_t i = 0; i < n; i++) {
        rng_total += h->t_rng_cnt[i]; seq_total += h->t_seq_bytes[i];
        B.rng_off.push_back(rng_total); B.seq_off.push_back(seq_total);
    }
    B.rng_q0.resize(rng_total); B.rng_len.resize(rng_total); B.rng_byte.resize(rng_total);
    B.seq.resize_uninit(seq_total);
    // (4) copy the kept bases
    h->pool->run(n_tasks, [&](int t) {
        const auto& kept = h->t_ranges[(size_t)t];
        size_t j = 0;
        while (j < kept.size()) {
            const size_t i = kept[j].rec, k = base + i;
            const svx_bam::RecRef& rr = refs[i];
            const uint8_t* sq = rr.r + 32 + rr.r[8] + 4 * (size_t)rd16(rr.r + 12);
            uint32_t slot = B.rng_off[k];                     // rng_off[k] = first range of record k (rng_off has a leading 0)
            uint64_t at = B.seq_off[k];
            for (; j < kept.size() && kept[j].rec == i; j++, slot++) {
                const uint32_t nb = ((uint32_t)kept[j].len + 1) / 2;
                memcpy(B.seq.data() + at, sq + kept[j].q0 / 2, nb);
                B.rng_q0[slot] = kept[j].q0; B.rng_len[slot] = kept[j].len; B.rng_byte[slot] = at;
                at += nb;
            }
        }
    });
    h->t_decode += now_s() - t0; t0 = now_s();
    for (size_t i = 0; i < n; i++) {
        if (i + 12 < n) h->names.prefetch(h->t_name_hash[i + 12]);
        B.read_id[base + i] = h->names.intern(h->t_name[i], h->t_name_len[i], h->t_name_hash[i]);
        B.sa_at[base + i] = B.sa_blob.size(); B.sa_len[base + i] = h->t_sa_len[i];
        if (h->t_sa_len[i]) B.sa_blob.append(h->t_sa[i], h->t_sa_len[i]);
    }
    h->pos = p;
    h->t_intern += now_s() - t0;
    return (int64_t)n;
}

// next record -> appended to the batch arrays; returns 1 = ok, 0 = EOF
static int parse_record(svx_bam* h) {
    if (!ensure(h, 4)) return 0;
    const uint32_t bs = rd32(h->buf.data() + h->pos);
    if (!ensure(h, 4 + (size_t)bs)) throw std::string("truncated BAM record");
    const uint8_t* r = h->buf.data() + h->pos + 4;
    const uint8_t* end = r + bs;
    h->pos += 4 + (size_t)bs;
    const int32_t tid = (int32_t)rd32(r), pos = (int32_t)rd32(r + 4);
    const unsigned l_name = r[8], mq = r[9];
    unsigned n_cig = rd16(r + 12); const unsigned flag = rd16(r + 14);
    const uint32_t l_seq = rd32(r + 16);
    const uint8_t* q = r + 32;
    const std::string name((const char*)q, l_name ? l_name - 1 : 0);
    q += l_name;
    const uint8_t* cig = q; q += 4 * (size_t)n_cig;
    const uint8_t* sq = q; q += (l_seq + 1) / 2;
    q += l_seq;
    const char* sa; size_t sa_n; const uint8_t* cg; uint32_t cg_n;
    scan_aux(q, end, sa, sa_n, cg, cg_n);
    if (cg && cg_n >= n_cig && cg_n < (1u << 29) && cg_placeholder(r, cig, n_cig, l_seq)) { cig = cg; n_cig = cg_n; }          // long CIGARs (> 65535 ops): see decode_run
    const int32_t rid = h->names.intern(name.data(), name.size(), name_hash(name.data(), name.size()));
    h->b->flag.push_back((uint16_t)(flag & 0x0fff)); h->b->tid.push_back(tid); h->b->bpos.push_back(pos); h->b->mapq.push_back((uint8_t)mq);
    h->b->lseq.push_back((int32_t)l_seq); h->b->read_id.push_back(rid);
    const size_t c0 = h->b->cigar.size();
    h->b->cigar.resize_uninit(c0 + n_cig);
    for (unsigned i = 0; i < n_cig; i++) h->b->cigar[c0 + i] = rd32(cig + 4 * (size_t)i);
    h->b->cigar_off.push_back(h->b->cigar.size());
    h->b->seq.append(sq, (l_seq + 1) / 2);
    h->b->seq_off.push_back(h->b->seq.size());
    h->b->sa_at.push_back(h->b->sa_blob.size()); h->b->sa_len.push_back((uint32_t)sa_n);
    if (sa_n) h->b->sa_blob.append(sa, sa_n);
    return 1;
}

static void clear_batch(svx_bam* h) {
    BatchArrays& B = *h->b;
    B.flag.clear(); B.tid.clear(); B.bpos.clear(); B.mapq.clear(); B.lseq.clear(); B.read_id.clear(); B.order.clear(); B.seg_order.clear();
    B.seg_off.clear(); B.cigar.clear(); B.seg_cigar.clear(); B.cigar_off.assign(1, 0); B.seq_off.assign(1, 0); B.seg_cigar_off.assign(1, 0);
    B.seq.clear(); B.seg_rev.clear(); B.seg_mapq.clear(); B.seg_tid.clear(); B.seg_pos.clear(); B.seg_lseq.clear();
    B.sa_blob.clear(); B.sa_at.clear(); B.sa_len.clear();
    B.rng_off.assign(1, 0); B.rng_q0.clear(); B.rng_len.clear(); B.rng_byte.clear();
}

// Read up to max_records records (query-name mode: never splits a read's group) and lay them out as an svx_batch whose
// arrays stay valid until the next call.  *n_out = 0 at end of file.  mode 0 = coordinate-sorted rules
// (src/svim/SVIM_COLLECT.py:132-167), 1 = query-name-sorted rules (:96-129).
// one chunk of the device reader: whole BGZF blocks up to dev_chunk_bytes of inflated data -> slot `slot` (runs on a background thread while the
// batches of the chunk before it are handed out; only this function advances dev_fpos)
static svx_bam::DevLoad dev_load_chunk(svx_bam* h, int slot, int carry_slot, uint64_t skip, int min_mapq, int mode, size_t budget_bytes, size_t budget_blocks) {
    svx_bam::DevLoad r;
    r.slot = slot; r.carry_slot = carry_slot; r.skip = skip; r.fpos_start = h->dev_fpos;
    std::vector<DevDecBlock> blocks;
    size_t total = 0;
    try {
        size_t fp = h->dev_fpos;
        std::swap(fp, h->fpos);                                   // (read_block walks h->fpos; the host reader is idle in device mode)
        while (total < budget_bytes && blocks.size() < budget_blocks) {
            RawBlock b;
            if (!read_block(h, b)) { r.file_done = true; break; }
            blocks.push_back(DevDecBlock{b.comp, (uint32_t)b.clen, b.isize, b.crc});
            total += b.isize;
        }
        std::swap(fp, h->fpos);
        h->dev_fpos = fp;
    } catch (const std::string& e) { r.rc = SVX_E_ARG; r.err = e; return r; }
    catch (const std::exception& e) { r.rc = SVX_E_ARG; r.err = e.what(); return r; }      // (this runs on a std::async thread: nothing may escape into future::get of a C entry point)
    if (blocks.empty() && carry_slot < 0) { r.file_done = true; r.empty = true; return r; }
    try {
        r.rc = devdec_load(h->dev, slot, blocks.data(), blocks.size(), carry_slot, skip, r.file_done, min_mapq, mode);
        if (r.rc == SVX_OK) r.rc = devdec_count(h->dev, slot, h->tid_limit, &r.n_rec, &r.n_valid);
        if (r.rc != SVX_OK) r.err = svx_last_error();
    } catch (const std::exception& e) { r.rc = SVX_E_ARG; r.err = e.what(); }
    return r;
}
static void dev_start_prefetch(svx_bam* h, int slot, int carry_slot, uint64_t skip, int min_mapq) {
    // budgets are fixed here, on the caller's thread (the loader runs beside the consumer)
    size_t bytes = h->dev_chunk_bytes, blocks = h->dev_chunk_blocks;
    if (h->dev_region_bytes && h->dev_region_bytes < bytes) bytes = h->dev_region_bytes;
    for (int g = 0; g < h->dev_grow; g++) { if (bytes < ((size_t)1 << 62)) bytes *= 2; if (blocks < ((size_t)1 << 62)) blocks *= 2; }
    if (h->dev_region_bytes && !h->dev_grow) h->dev_region_bytes = std::min(h->dev_region_bytes * 4, h->dev_chunk_bytes);
    h->dev_last_slot = slot;
    h->dev_future = std::async(std::launch::async, dev_load_chunk, h, slot, carry_slot, skip, min_mapq, h->dev_mode, bytes, blocks);
    h->dev_prefetching = true;
}
// the slot after the one loaded last - never restarted at 0: see dev_last_slot
static int dev_next_slot(const svx_bam* h) { return h->dev_last_slot < 0 ? 0 : (h->dev_last_slot + 1) % 3; }
static void dev_drop_prefetch(svx_bam* h) {
    if (h->dev_prefetching) { (void)h->dev_future.get(); h->dev_prefetching = false; }
    h->dev_last_slot = h->dev_handed_slot;                // whatever was loaded beyond the last handed-out chunk is discarded: the rotation goes on behind that chunk
}

// device decode: batches are views of the current chunk's arrays; the next chunk is inflated and decoded meanwhile
static int read_batch_device(svx_bam* h, int64_t max_records, int mode, int min_mapq, svx_batch* out, int64_t* n_out) {
    *n_out = 0;
    memset(out, 0, sizeof *out);
    if (h->dev_cur < 0 && !h->dev_prefetching) h->dev_mode = mode;           // a fresh start (open, rewind, seek): the mode of this pass
    else if (mode != h->dev_mode) return bam_fail(SVX_E_ARG, "the sort mode of a pass over the file cannot change between batches (rewind first)");
    for (;;) {
        if (h->dev_cur >= 0 && h->dev_first < h->dev_valid) {
            int64_t count = std::min<int64_t>(max_records, h->dev_valid - h->dev_first);
            const int rc = devdec_batch(h->dev, h->dev_cur, h->dev_first, &count, mode, min_mapq, out);      // (query-name mode: grows to the end of the read's group)
            if (rc != SVX_OK) return rc;
            h->dev_first += count; h->total_records += count; *n_out = count;
            h->dev_handed_slot = h->dev_cur;
            return SVX_OK;
        }
        if (h->dev_region_done || h->dev_file_done) return SVX_OK;          // end of the region / of the file
        if (!h->dev_prefetching) dev_start_prefetch(h, dev_next_slot(h), h->dev_have_carry ? h->dev_cur : -1, h->dev_skip, min_mapq);
        svx_bam::DevLoad r = h->dev_future.get();
        h->dev_prefetching = false;
        h->dev_skip = 0;
        if (r.rc != SVX_OK) return bam_fail(r.rc, r.err);
        if (r.empty) { h->dev_file_done = true; return SVX_OK; }
        if (r.n_rec == 0 && !r.file_done) {
            // not one complete record in the chunk (a header or a record longer than the chunk: only with the small chunks of the test hooks).  The rotation
            // must not advance for it - the slot after this one may still be in use - so the SAME slot is loaded again from the same place with twice the budget
            if (h->dev_grow >= 24) return bam_fail(SVX_E_ARG, "device BAM decode: no complete record within the largest chunk");
            h->dev_grow++;
            h->dev_fpos = r.fpos_start;
            dev_start_prefetch(h, r.slot, r.carry_slot, r.skip, min_mapq);
            continue;
        }
        h->dev_grow = 0;
        h->dev_cur = r.slot; h->dev_first = 0; h->dev_valid = r.n_valid; h->dev_have_carry = true;
        h->dev_file_done = r.file_done;                                     // (the records of this last chunk are still to be handed out: checked after them)
        if (r.n_valid < r.n_rec) h->dev_region_done = true;                 // contig-range reading: the range ends where the next contig (or the unplaced tail) begins
        // (a chunk without one complete record is carried over whole into the next load; devdec_load refuses when a record outgrows its carry-over room)
        if (!h->dev_file_done && !h->dev_region_done) dev_start_prefetch(h, dev_next_slot(h), h->dev_cur, 0, min_mapq);
    }
}

extern "C" int svx_bam_set_device_decode(svx_bam* h, int device) {
    if (!h) return bam_fail(SVX_E_ARG, "null handle");
    if (h->dev) { dev_drop_prefetch(h); devdec_destroy(h->dev); h->dev = nullptr; }
    if (device < 0) return SVX_OK;
    const int rc = devdec_create(device, granted_cpus(), (int32_t)h->ref_names.size(), h->ref_len.data(), h->names_blob.c_str(), h->contig_rank.data(), &h->dev);
    if (rc != SVX_OK) { h->dev = nullptr; return rc; }
    h->dev_device = device;
    { const char* e = getenv("SVX_BAM_DEV_CHUNK_MB"); if (e && atoll(e) > 0) h->dev_chunk_bytes = (size_t)atoll(e) << 20; }
    { const char* e = getenv("SVX_BAM_DEV_CHUNK_BLOCKS"); if (e && atoll(e) > 0) h->dev_chunk_blocks = (size_t)atoll(e); }
    if (h->prefetch_active) { h->prefetch.wait(); h->prefetch_active = false; }
    h->dev_fpos = 0; h->dev_skip = h->header_bytes; h->dev_file_done = false; h->dev_region_done = false; h->dev_cur = -1; h->dev_first = h->dev_valid = 0; h->dev_have_carry = false;
    h->dev_last_slot = h->dev_handed_slot = -1; h->dev_grow = 0; h->dev_region_bytes = 0;                      // a fresh decoder: nothing handed out from its slots yet
    return SVX_OK;
}

extern "C" int svx_bam_read_batch(svx_bam* h, int64_t max_records, int mode, int min_mapq, svx_batch* out, int64_t* n_out) {
    if (h->dev) return read_batch_device(h, max_records, mode == 1 ? 1 : 0, min_mapq, out, n_out);
    BatchArrays* const handed_out = h->b;                                                 // the set the previous call handed out: the caller may still upload from it
    try {
        h->b = &h->ba[h->b == &h->ba[0] ? 1 : 0];                                        // the previous batch stays valid during this read
        clear_batch(h);
        const bool sparse = h->seq_min_ins > 0 && mode == 0;
        int64_t n = 0;
        while (n < max_records && !h->region_done) {
            if (!ensure(h, 4)) break;                                                   // end of file
            const uint32_t bs = rd32(h->buf.data() + h->pos);
            if (!ensure(h, 4 + (size_t)bs)) throw std::string("truncated BAM record");
            const int64_t got = decode_run(h, max_records - n, sparse);                  // at least the record just made available
            if (got == 0 && !h->region_done) throw std::string("BAM record larger than the inflate window");
            n += got;
        }
        if (mode == 1 && n == max_records) {
            // finish the current read group: keep reading while the name does not change (peek = parse, names are interned)
            for (;;) {
                if (!ensure(h, 4)) break;
                const uint32_t bs = rd32(h->buf.data() + h->pos);
                if (!ensure(h, 4 + (size_t)bs)) throw std::string("truncated BAM record");
                const uint8_t* r = h->buf.data() + h->pos + 4;
                const std::string name((const char*)r + 32, r[8] ? r[8] - 1 : 0);
                const int32_t known = h->names.find(name.data(), name.size(), name_hash(name.data(), name.size()));
                // a record of another read stays where it is: nothing has advanced h->pos (ensure() may have MOVED the window - the position from before it
                // is stale then; restoring it, as this loop once did, sent the next batch into the middle of a record whenever the peek crossed a chunk end)
                if (known < 0 || known != h->b->read_id[(size_t)n - 1]) break;
                parse_record(h); n++;
            }
        }
        *n_out = n;
        if (n == 0) {
            // end of a region / of the file: nothing is handed out, so the sets must NOT alternate - the next read (after svx_bam_seek to another
            // region) has to fill THIS set again, not the one that still holds the last batch of the region that just ended
            memset(out, 0, sizeof *out);
            h->b = handed_out;
            return SVX_OK;
        }
        h->b->order.assign((size_t)n, 0); h->b->seg_order.assign((size_t)n, 0); h->b->seg_off.assign((size_t)n + 1, 0);
        if (mode == 0) {
            for (int64_t i = 0; i < n; i++) {
                h->b->order[(size_t)i] = (uint32_t)(2 * i); h->b->seg_order[(size_t)i] = (uint32_t)(2 * i + 1);
                h->b->seg_off[(size_t)i] = (uint32_t)h->b->seg_tid.size();
                const unsigned f = h->b->flag[(size_t)i];
                if ((f & (4u | 256u | 2048u)) || (int)h->b->mapq[(size_t)i] < min_mapq || !h->b->sa_len[(size_t)i]) continue;
                h->b->flag[(size_t)i] |= SVX_FLAG_SA;
                const int rc = append_sa(h, h->b->sa_blob.data() + h->b->sa_at[(size_t)i], h->b->sa_len[(size_t)i], h->b->lseq[(size_t)i]);
                if (rc != SVX_OK) return rc;
            }
            h->b->seg_off[(size_t)n] = (uint32_t)h->b->seg_tid.size();
        } else {
            uint32_t slot = 0;
            int64_t i = 0;
            while (i < n) {
                int64_t j = i;
                while (j < n && h->b->read_id[(size_t)j] == h->b->read_id[(size_t)i]) j++;
                std::vector<int64_t> prim, sup;
                for (int64_t k = i; k < j; k++) { const unsigned f = h->b->flag[(size_t)k]; if (f & 256u) continue; if (f & 2048u) sup.push_back(k); else prim.push_back(k); }
                const bool ok = prim.size() == 1 && !(h->b->flag[(size_t)prim[0]] & 4u) && (int)h->b->mapq[(size_t)prim[0]] >= min_mapq;
                for (int64_t k = i; k < j; k++) { h->b->flag[(size_t)k] |= SVX_FLAG_SKIP; h->b->seg_off[(size_t)k] = (uint32_t)h->b->seg_tid.size(); }
                if (ok) {
                    const int64_t p = prim[0];
                    // segment rows belong to the primary: the offsets of the records after it inside the group move past them
                    std::vector<int64_t> good;
                    for (int64_t k : sup) if (!(h->b->flag[(size_t)k] & 4u) && (int)h->b->mapq[(size_t)k] >= min_mapq) good.push_back(k);
                    h->b->flag[(size_t)p] &= (uint16_t)~SVX_FLAG_SKIP;
                    h->b->order[(size_t)p] = slot;
                    for (int64_t k = i; k <= p; k++) h->b->seg_off[(size_t)k] = (uint32_t)h->b->seg_tid.size();
                    for (size_t qn = 0; qn < good.size(); qn++) {
                        const int64_t k = good[qn];
                        h->b->flag[(size_t)k] &= (uint16_t)~SVX_FLAG_SKIP;
                        h->b->order[(size_t)k] = slot + 1 + (uint32_t)qn;
                        h->b->seg_tid.push_back(h->b->tid[(size_t)k]); h->b->seg_pos.push_back(h->b->bpos[(size_t)k]);
                        h->b->seg_rev.push_back((h->b->flag[(size_t)k] & 16u) ? 1 : 0); h->b->seg_mapq.push_back(h->b->mapq[(size_t)k]);
                        h->b->seg_lseq.push_back(h->b->lseq[(size_t)k]);
                        for (uint64_t c = h->b->cigar_off[(size_t)k]; c < h->b->cigar_off[(size_t)k + 1]; c++) h->b->seg_cigar.push_back(h->b->cigar[(size_t)c]);
                        h->b->seg_cigar_off.push_back(h->b->seg_cigar.size());
                    }
                    for (int64_t k = p + 1; k < j; k++) h->b->seg_off[(size_t)k] = (uint32_t)h->b->seg_tid.size();
                    h->b->seg_order[(size_t)p] = slot + 1 + (uint32_t)good.size();
                    slot += (uint32_t)good.size() + 2;
                }
                i = j;
            }
            h->b->seg_off[(size_t)n] = (uint32_t)h->b->seg_tid.size();
        }
        // never hand out null pointers for empty arrays
        auto pad = [](auto& v) { if (v.empty()) v.resize(1); };
        if (h->b->cigar.empty()) { h->b->cigar.reserve(1); h->b->cigar[0] = 0; }
        if (h->b->seq.empty()) { h->b->seq.reserve(1); h->b->seq[0] = 0; }
        pad(h->b->seg_tid); pad(h->b->seg_pos); pad(h->b->seg_rev); pad(h->b->seg_mapq); pad(h->b->seg_lseq); pad(h->b->seg_cigar);
        pad(h->b->flag); pad(h->b->tid); pad(h->b->bpos); pad(h->b->mapq); pad(h->b->lseq); pad(h->b->read_id); pad(h->b->order); pad(h->b->seg_order);
        memset(out, 0, sizeof *out);
        out->on_device = 0; out->n_rec = n; out->flag = h->b->flag.data(); out->tid = h->b->tid.data(); out->pos = h->b->bpos.data(); out->mapq = h->b->mapq.data();
        out->lseq = h->b->lseq.data(); out->read_id = h->b->read_id.data(); out->order = h->b->order.data(); out->seg_order = h->b->seg_order.data();
        out->cigar_off = h->b->cigar_off.data(); out->cigar = h->b->cigar.data(); out->seq_off = h->b->seq_off.data(); out->seq = h->b->seq.data();
        out->seg_off = h->b->seg_off.data(); out->n_seg = (int64_t)h->b->seg_cigar_off.size() - 1; out->seg_tid = h->b->seg_tid.data(); out->seg_pos = h->b->seg_pos.data();
        out->seg_rev = h->b->seg_rev.data(); out->seg_mapq = h->b->seg_mapq.data(); out->seg_lseq = h->b->seg_lseq.data(); out->seg_cigar_off = h->b->seg_cigar_off.data();
        out->seg_cigar = h->b->seg_cigar.data(); out->n_contig = (int32_t)h->ref_names.size(); out->contig_rank = h->contig_rank.data();
        if (sparse) {
            pad(h->b->rng_q0); pad(h->b->rng_len); pad(h->b->rng_byte);
            out->seq_rng_off = h->b->rng_off.data(); out->seq_rng_q0 = h->b->rng_q0.data(); out->seq_rng_len = h->b->rng_len.data();
            out->seq_rng_byte = h->b->rng_byte.data(); out->n_seq_rng = (int64_t)h->b->rng_off[(size_t)n];
        }
        h->total_records += n;
    } catch (const std::string& e) { h->b = handed_out; return bam_fail(SVX_E_ARG, e); }
      catch (const std::exception& e) { h->b = handed_out; return bam_fail(SVX_E_ARG, std::string("reader: ") + e.what()); }
    return SVX_OK;
}

// read names interned so far: NUL-separated blob in id order
extern "C" int svx_bam_read_names(svx_bam* h, int64_t* n_names, const char** blob, int64_t* blob_len) {
    if (h->dev) {
        if (h->dev_prefetching) h->dev_future.wait();          // (the loader appends the names of the chunk it decodes)
        const std::vector<std::string>& nm = devdec_names(h->dev);
        h->dev_names_blob.clear();
        for (const auto& x : nm) { h->dev_names_blob += x; h->dev_names_blob.push_back('\0'); }
        *n_names = (int64_t)nm.size(); *blob = h->dev_names_blob.data(); *blob_len = (int64_t)h->dev_names_blob.size();
        return SVX_OK;
    }
    *n_names = (int64_t)h->names.size(); *blob = h->names.blob.data(); *blob_len = (int64_t)h->names.blob.size();
    return SVX_OK;
}
