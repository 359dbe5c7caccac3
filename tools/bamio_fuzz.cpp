// Host reader (svim_amd/csrc/bamio.cpp) on damaged BAM files, built with AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_host_cpu.py):
//   bamio_fuzz <seed.bam> <iterations>
// The seed file's BGZF blocks are inflated, the RECORD STREAM is damaged (bytes overwritten, fields of record headers set to extreme values, the stream cut),
// deflated again into well-formed BGZF blocks with correct CRCs - so the damage reaches the record decoder - and read to the end in both sort modes, with and
// without the sparse-SEQ filter; every other iteration the damage is done to the compressed file instead.  A read may fail (error code) or succeed; it must not
// touch memory outside its buffers.  The device-side entry points bamio.cpp links against are stubs here (the host reader never calls them).
// Build: g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined svim_amd/csrc/bamio.cpp tools/bamio_fuzz.cpp -lz -lpthread -o bamio_fuzz
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/svx.h"
#include "../svim_amd/csrc/devdec.hpp"

thread_local std::string g_svx_err;
extern "C" const char* svx_last_error(void) { return g_svx_err.c_str(); }
// device-side entry points (bgzf.hip, bamdev.hip): never reached by the host reader
extern "C" int svx_inflater_create(int, svx_inflater**) { return SVX_E_NODEVICE; }
extern "C" void svx_inflater_destroy(svx_inflater*) {}
extern "C" void* svx_inflater_staging(svx_inflater*, int, uint64_t) { return nullptr; }
extern "C" int svx_inflater_enqueue(svx_inflater*, int, int64_t, const uint64_t*, const uint32_t*, const uint32_t*, const uint64_t*, uint64_t, uint8_t*, uint64_t, int) { return SVX_E_NODEVICE; }
extern "C" int svx_inflater_wait(svx_inflater*, int, float*) { return SVX_E_NODEVICE; }
// ---- a CPU stand-in for the device decoder (bamdev.hip) with the same contract towards bamio.cpp: chunk slots, the unconsumed tail of one slot carried into the
// next load, query-name mode holding the last read group back, batches as views of a slot's arrays.  What it is for: the HOST side of the device reader (slot
// rotation across seek / rewind, grow-and-retry of chunks without a complete record, the chunk budget of contig ranges) under the sanitizers - every array of a slot
// is heap memory that is FREED when the slot is loaded again, so a batch that is read after its slot was reused is a use-after-free AddressSanitizer reports.
struct MockSlot {
    std::vector<uint8_t>* stream = nullptr; uint64_t tail_start = 0; int64_t n_rec = 0;
    uint16_t* flag = nullptr; int32_t* tid = nullptr; int32_t* pos = nullptr; uint8_t* mapq = nullptr; int32_t* lseq = nullptr; int32_t* read_id = nullptr;
    uint32_t* order = nullptr; uint64_t* cigar_off = nullptr; uint32_t* cigar = nullptr; uint64_t* seq_off = nullptr; uint32_t* seg_off = nullptr; uint64_t* seg_cigar_off = nullptr;
    void release() { delete stream; stream = nullptr; delete[] flag; delete[] tid; delete[] pos; delete[] mapq; delete[] lseq; delete[] read_id; delete[] order; delete[] cigar_off;
                     delete[] cigar; delete[] seq_off; delete[] seg_off; delete[] seg_cigar_off;
                     flag = nullptr; tid = pos = lseq = read_id = nullptr; mapq = nullptr; order = seg_off = cigar = nullptr; cigar_off = seq_off = seg_cigar_off = nullptr; n_rec = 0; tail_start = 0; }
};
struct svx_devdec { MockSlot slot[3]; std::vector<std::string> names; std::vector<int32_t> rank; long long loads = 0; };
static uint32_t m32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
int devdec_create(int, int, int32_t n_ref, const int32_t*, const char*, const int32_t* contig_rank, svx_devdec** out) {
    svx_devdec* d = new svx_devdec(); d->rank.assign(contig_rank, contig_rank + n_ref); *out = d; return SVX_OK;
}
void devdec_destroy(svx_devdec* d) { if (!d) return; for (auto& s : d->slot) s.release(); delete d; }
int devdec_load(svx_devdec* d, int slot, const DevDecBlock* blocks, size_t n, int carry_slot, uint64_t skip, bool final_chunk, int, int mode) {
    std::vector<uint8_t>* st = new std::vector<uint8_t>();
    if (carry_slot >= 0) { const MockSlot& c = d->slot[carry_slot]; if (!c.stream) { delete st; g_svx_err = "mock: carry from an empty slot"; return SVX_E_STATE; }
                           st->assign(c.stream->begin() + (long)c.tail_start, c.stream->end()); }
    const size_t carried = st->size();
    for (size_t b = 0; b < n; b++) {
        if (blocks[b].isize == 0) continue;                     // (the EOF block)
        const size_t o = st->size(); st->resize(o + blocks[b].isize);
        z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
        zs.next_in = const_cast<Bytef*>(blocks[b].comp); zs.avail_in = blocks[b].clen; zs.next_out = st->data() + o; zs.avail_out = blocks[b].isize;
        const int rc = inflate(&zs, Z_FINISH); const bool ok = rc == Z_STREAM_END && zs.total_out == blocks[b].isize; inflateEnd(&zs);
        if (!ok || (uint32_t)crc32(crc32(0, nullptr, 0), st->data() + o, blocks[b].isize) != blocks[b].crc) { delete st; g_svx_err = "mock: damaged BGZF block"; return SVX_E_ARG; }
    }
    d->slot[slot].release();                                   // (after the carry was copied: slot may equal carry_slot when a chunk is loaded again with a larger budget)
    d->loads++;
    MockSlot& S = d->slot[slot];
    S.stream = st;
    std::vector<uint64_t> rec;
    uint64_t p = carried ? 0 : skip;
    if (carried && skip) { delete st; S.stream = nullptr; g_svx_err = "mock: skip together with a carry"; return SVX_E_STATE; }
    while (p + 4 <= st->size()) { const uint64_t nx = p + 4ull + m32(st->data() + p); if (nx > st->size()) break; if (nx - p < 36) { g_svx_err = "mock: corrupt record"; return SVX_E_ARG; } rec.push_back(p); p = nx; }
    if (final_chunk && p != st->size()) { g_svx_err = "mock: truncated record at the end of the file"; return SVX_E_ARG; }
    auto name_of = [&](uint64_t r) { const uint8_t* q = st->data() + r + 4; return std::string((const char*)q + 32, q[8] ? q[8] - 1 : 0); };
    size_t keep = rec.size();
    if (mode == 1 && !final_chunk && keep) { const std::string last = name_of(rec[keep - 1]); while (keep > 0 && name_of(rec[keep - 1]) == last) keep--; }     // the last group waits for the next load
    S.tail_start = keep < rec.size() ? rec[keep] : p;
    S.n_rec = (int64_t)keep;
    const size_t m = keep;
    S.flag = new uint16_t[m + 1]; S.tid = new int32_t[m + 1]; S.pos = new int32_t[m + 1]; S.mapq = new uint8_t[m + 1]; S.lseq = new int32_t[m + 1]; S.read_id = new int32_t[m + 1];
    S.order = new uint32_t[m + 1]; S.cigar_off = new uint64_t[m + 1]; S.seq_off = new uint64_t[m + 1]; S.seg_off = new uint32_t[m + 1](); S.seg_cigar_off = new uint64_t[1]();
    uint64_t nc = 0;
    for (size_t i = 0; i < m; i++) nc += st->data()[rec[i] + 4 + 12] | (st->data()[rec[i] + 4 + 13] << 8);
    S.cigar = new uint32_t[nc + 1];
    uint64_t c = 0;
    for (size_t i = 0; i < m; i++) {
        const uint8_t* q = st->data() + rec[i] + 4;
        S.tid[i] = (int32_t)m32(q); S.pos[i] = (int32_t)m32(q + 4); S.mapq[i] = q[9]; S.flag[i] = (uint16_t)((q[14] | (q[15] << 8)) & 0x0fff); S.lseq[i] = (int32_t)m32(q + 16);
        S.order[i] = (uint32_t)(2 * i); S.cigar_off[i] = c; S.seq_off[i] = rec[i] + 4 + 32 + q[8] + 4ull * (q[12] | (q[13] << 8));
        const unsigned ncig = q[12] | (q[13] << 8);
        for (unsigned k = 0; k < ncig; k++) S.cigar[c++] = m32(q + 32 + q[8] + 4 * k);
        const std::string nm = name_of(rec[i]);
        int32_t id = -1;
        for (size_t k = d->names.size(); k-- > 0 && d->names.size() - k < 64;) if (d->names[k] == nm) { id = (int32_t)k; break; }     // (names repeat next to each other)
        if (id < 0) { id = (int32_t)d->names.size(); d->names.push_back(nm); }
        S.read_id[i] = id;
    }
    S.cigar_off[m] = c; S.seq_off[m] = 0;
    return SVX_OK;
}
int devdec_count(svx_devdec* d, int slot, int32_t tid_limit, int64_t* n_rec, int64_t* n_valid) {
    const MockSlot& S = d->slot[slot];
    *n_rec = S.n_rec; int64_t v = S.n_rec;
    if (tid_limit != -2) for (int64_t i = 0; i < S.n_rec; i++) if (S.tid[i] < 0 || S.tid[i] > tid_limit) { v = i; break; }
    *n_valid = v;
    return SVX_OK;
}
int devdec_batch(svx_devdec* d, int slot, int64_t first, int64_t* count, int mode, int, svx_batch* out) {
    const MockSlot& S = d->slot[slot];
    if (!S.stream || first < 0 || first + *count > S.n_rec) { g_svx_err = "mock: batch outside its slot"; return SVX_E_STATE; }
    if (mode == 1) while (first + *count < S.n_rec && S.read_id[first + *count] == S.read_id[first + *count - 1]) (*count)++;
    memset(out, 0, sizeof *out);
    out->on_device = 1; out->n_rec = *count;
    out->flag = S.flag + first; out->tid = S.tid + first; out->pos = S.pos + first; out->mapq = S.mapq + first; out->lseq = S.lseq + first; out->read_id = S.read_id + first;
    out->order = S.order + first; out->seg_order = S.order + first; out->cigar_off = S.cigar_off + first; out->cigar = S.cigar; out->seq_off = S.seq_off + first; out->seq = S.stream->data();
    out->seg_off = S.seg_off + first; out->seg_cigar_off = S.seg_cigar_off; out->n_contig = (int32_t)d->rank.size(); out->contig_rank = d->rank.data();
    return SVX_OK;
}
const std::vector<std::string>& devdec_names(svx_devdec* d) { return d->names; }
void devdec_stats(svx_devdec*, DevDecStats*) {}
void devdec_reset_names(svx_devdec* d) { d->names.clear(); }

static std::vector<uint8_t> read_file(const char* p) {
    std::vector<uint8_t> d; FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(2); }
    uint8_t buf[1 << 16]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + k); fclose(f); return d;
}
// every BGZF block of a file image inflated into one stream
static std::vector<uint8_t> inflate_all(const std::vector<uint8_t>& file) {
    std::vector<uint8_t> out; size_t at = 0;
    while (at + 18 <= file.size()) {
        const uint8_t* hd = file.data() + at;
        const size_t blen = (size_t)(hd[16] | (hd[17] << 8)) + 1;
        if (at + blen > file.size()) break;
        const uint32_t isize = hd[blen - 4] | (hd[blen - 3] << 8) | (hd[blen - 2] << 16) | ((uint32_t)hd[blen - 1] << 24);
        const size_t o = out.size(); out.resize(o + isize);
        z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15);
        zs.next_in = const_cast<Bytef*>(hd + 18); zs.avail_in = (uInt)(blen - 26); zs.next_out = out.data() + o; zs.avail_out = isize;
        inflate(&zs, Z_FINISH); inflateEnd(&zs);
        at += blen;
    }
    return out;
}
// a stream cut into well-formed BGZF blocks (+ the EOF block)
static std::vector<uint8_t> bgzf_all(const std::vector<uint8_t>& raw, size_t block) {
    std::vector<uint8_t> out;
    auto put = [&](const uint8_t* p, size_t n) {
        std::vector<uint8_t> c(n + n / 8 + 256);
        z_stream zs; memset(&zs, 0, sizeof zs); deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<Bytef*>(p); zs.avail_in = (uInt)n; zs.next_out = c.data(); zs.avail_out = (uInt)c.size();
        deflate(&zs, Z_FINISH); const size_t cl = zs.total_out; deflateEnd(&zs);
        const size_t bs = cl + 25;
        const uint8_t hd[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bs & 255), (uint8_t)(bs >> 8)};
        out.insert(out.end(), hd, hd + 18); out.insert(out.end(), c.begin(), c.begin() + (long)cl);
        const uint32_t crc = (uint32_t)crc32(crc32(0, nullptr, 0), p, (uInt)n), is = (uint32_t)n;
        for (int k = 0; k < 4; k++) out.push_back((uint8_t)(crc >> (8 * k)));
        for (int k = 0; k < 4; k++) out.push_back((uint8_t)(is >> (8 * k)));
    };
    for (size_t at = 0; at < raw.size(); at += block) put(raw.data() + at, raw.size() - at < block ? raw.size() - at : block);
    put(nullptr, 0);
    return out;
}
static long long read_to_end(const char* path, int mode, int seq_filter, int threads) {
    svx_bam* h = nullptr;
    if (svx_bam_open(path, threads, &h) != SVX_OK) return -1;
    if (seq_filter) (void)svx_bam_set_seq_filter(h, 40);
    long long total = 0;
    for (int guard = 0; guard < 100000; guard++) {
        svx_batch b; int64_t n = 0;
        if (svx_bam_read_batch(h, 97, mode, 20, &b, &n) != SVX_OK) { total = -2 - total; break; }
        if (n == 0) break;
        // touch what was handed out: the offsets must describe memory the batch owns
        unsigned long long sum = 0;
        for (int64_t i = 0; i < n; i++) {
            sum += b.flag[i] + (unsigned)b.tid[i] + (unsigned)b.pos[i] + b.mapq[i] + (unsigned)b.lseq[i];
            for (uint64_t k = b.cigar_off[i]; k < b.cigar_off[i + 1]; k++) sum += b.cigar[k];
            for (uint32_t s = b.seg_off[i]; s < b.seg_off[i + 1]; s++) { sum += (unsigned)b.seg_pos[s]; for (uint64_t k = b.seg_cigar_off[s]; k < b.seg_cigar_off[s + 1]; k++) sum += b.seg_cigar[k]; }
            if (!b.seq_rng_off && b.seq) for (uint64_t k = b.seq_off[i]; k < b.seq_off[i] + ((uint64_t)b.lseq[i] + 1) / 2; k++) sum += b.seq[k];
        }
        if (sum == 0x123456789abcdefull) printf("!");
        total += n;
    }
    int64_t nn = 0, bl = 0; const char* names = nullptr;
    (void)svx_bam_read_names(h, &nn, &names, &bl);
    svx_bam_close(h);
    return total;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: bamio_fuzz <seed.bam> <iterations>\n"); return 2; }
    const std::vector<uint8_t> file = read_file(argv[1]);
    const std::vector<uint8_t> raw = inflate_all(file);
    const int iters = atoi(argv[2]);
    const std::string tmp = std::string(argv[1]) + ".fuzz.bam";
    unsigned long long x = 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    // offsets of the records in the raw stream (header: magic, l_text, text, n_ref, refs)
    std::vector<size_t> rec;
    {
        size_t p = 8 + (raw[4] | (raw[5] << 8) | (raw[6] << 16) | ((size_t)raw[7] << 24));
        const uint32_t n_ref = raw[p] | (raw[p + 1] << 8) | (raw[p + 2] << 16) | ((uint32_t)raw[p + 3] << 24); p += 4;
        for (uint32_t r = 0; r < n_ref; r++) { const uint32_t l = raw[p] | (raw[p + 1] << 8) | (raw[p + 2] << 16) | ((uint32_t)raw[p + 3] << 24); p += 8 + l; }
        while (p + 4 <= raw.size()) { rec.push_back(p); p += 4 + (raw[p] | (raw[p + 1] << 8) | (raw[p + 2] << 16) | ((size_t)raw[p + 3] << 24)); }
    }
    const long long clean = read_to_end(argv[1], 0, 0, 2);
    printf("seed: %zu records in the stream, %lld read\n", rec.size(), clean);
    if (iters == 0) {
        // the undamaged file with 1..6 threads, a rewind in the middle of a pass, small batches: what ThreadSanitizer is pointed at (-fsanitize=thread)
        long long bad = 0;
        for (int threads = 1; threads <= 6; threads++) {
            svx_bam* h = nullptr;
            if (svx_bam_open(argv[1], threads, &h) != SVX_OK) { bad++; continue; }
            if (threads & 1) (void)svx_bam_set_seq_filter(h, 40);
            long long total = 0;
            for (int pass = 0; pass < 3; pass++) {
                total = 0;
                for (int k = 0; ; k++) {
                    svx_batch b; int64_t n = 0;
                    if (svx_bam_read_batch(h, 53 + 40 * pass, pass == 2 ? 1 : 0, 20, &b, &n) != SVX_OK) { bad++; fprintf(stderr, "threads %d pass %d batch %d: %s\n", threads, pass, k, svx_last_error()); break; }
                    if (n == 0) break;
                    total += n;
                    if (pass == 0 && k == 3) break;                   // leave in the middle: the prefetch of the next chunk is in flight
                }
                if (svx_bam_rewind(h) != SVX_OK) bad++;
            }
            if (total != clean) { bad++; fprintf(stderr, "threads %d: %lld records in the last pass, %lld expected\n", threads, total, clean); }
            svx_bam_close(h);
        }
        printf("threads 1..6 with rewinds: %lld problems\n", bad);
        // contig-range reading (svx_bam_seek): from the first record of every reference id, in an order that is not the file's, twice, each range read to
        // its end - the count must be the number of records with that id; virtual offsets from the block layout of the file
        std::vector<size_t> blk_file, blk_raw;                     // file offset / offset in the inflated stream of every BGZF block
        { size_t at = 0, ro = 0; while (at + 18 <= file.size()) { const size_t blen = (size_t)(file[at + 16] | (file[at + 17] << 8)) + 1; if (at + blen > file.size()) break;
              blk_file.push_back(at); blk_raw.push_back(ro); ro += file[at + blen - 4] | (file[at + blen - 3] << 8) | (file[at + blen - 2] << 16) | ((size_t)file[at + blen - 1] << 24); at += blen; } }
        auto voff_of = [&](size_t raw_off) { size_t b = 0; while (b + 1 < blk_raw.size() && blk_raw[b + 1] <= raw_off) b++; return ((unsigned long long)blk_file[b] << 16) | (unsigned long long)(raw_off - blk_raw[b]); };
        std::vector<int32_t> tids; std::vector<size_t> first_of; std::vector<long long> count_of;
        for (size_t p : rec) { int32_t t; memcpy(&t, raw.data() + p + 4, 4); if (t < 0) continue;
            if (tids.empty() || tids.back() != t) { tids.push_back(t); first_of.push_back(p); count_of.push_back(0); } count_of.back()++; }
        long long rbad = 0;
        for (int threads = 1; threads <= 3; threads++) {
            svx_bam* h = nullptr;
            if (svx_bam_open(argv[1], threads, &h) != SVX_OK) { rbad++; continue; }
            for (int round = 0; round < 2; round++)
                for (size_t k = 0; k < tids.size(); k++) {
                    const size_t r = (k * 7 + 3 + (size_t)round) % tids.size();
                    if (svx_bam_seek(h, voff_of(first_of[r]), tids[r]) != SVX_OK) { rbad++; continue; }
                    long long got = 0;
                    for (;;) { svx_batch b; int64_t n = 0; if (svx_bam_read_batch(h, 61, 0, 20, &b, &n) != SVX_OK) { rbad++; fprintf(stderr, "region tid %d: %s\n", tids[r], svx_last_error()); break; } if (n == 0) break; got += n; }
                    if (got != count_of[r]) { rbad++; fprintf(stderr, "region tid %d: %lld records, %lld expected\n", tids[r], got, count_of[r]); }
                }
            svx_bam_close(h);
        }
        printf("contig ranges out of file order: %lld problems\n", rbad);
        // ---- the device reader's host side over the CPU stand-in of the decoder (above): chunks of 1 .. all blocks, both modes, rewinds, contig ranges out of
        // file order; the batch handed out before is read again AFTER the next read (a slot reused too early = use-after-free)
        struct Want { int32_t tid, pos, lseq; uint16_t flag; uint32_t ncig; std::string name; };
        std::vector<Want> want;
        for (size_t p : rec) { const uint8_t* q = raw.data() + p + 4; Want w; w.tid = (int32_t)m32(q); w.pos = (int32_t)m32(q + 4); w.lseq = (int32_t)m32(q + 16);
            w.flag = (uint16_t)((q[14] | (q[15] << 8)) & 0x0fff); w.ncig = q[12] | (q[13] << 8); w.name.assign((const char*)q + 32, q[8] ? q[8] - 1 : 0); want.push_back(w); }
        std::vector<size_t> first_idx(tids.size(), 0);
        { size_t k = 0; for (size_t i = 0; i < rec.size() && k < tids.size(); i++) if (rec[i] == first_of[k]) { first_idx[k] = i; k++; } }
        long long dbad = 0;
        auto touch = [](const svx_batch& b) { unsigned long long s = 0; for (int64_t i = 0; i < b.n_rec; i++) { s += b.flag[i] + (unsigned)b.tid[i] + (unsigned)b.read_id[i];
                                                                for (uint64_t k = b.cigar_off[i]; k < b.cigar_off[i + 1]; k++) s += b.cigar[k]; } return s; };
        const char* chunk_blocks[5] = {"1", "2", "3", "7", "1000000"};
        for (int cb = 0; cb < 5; cb++) {
            setenv("SVX_BAM_DEV_CHUNK_BLOCKS", chunk_blocks[cb], 1);
            svx_bam* h = nullptr;
            if (svx_bam_open(argv[1], 2, &h) != SVX_OK || svx_bam_set_device_decode(h, 0) != SVX_OK) { dbad++; fprintf(stderr, "device mode: %s\n", svx_last_error()); continue; }
            auto names_now = [&]() { std::vector<std::string> v; int64_t nn = 0, bl = 0; const char* blob = nullptr; (void)svx_bam_read_names(h, &nn, &blob, &bl);
                                     const char* q = blob; for (int64_t i = 0; i < nn; i++) { v.emplace_back(q); q += v.back().size() + 1; } return v; };
            // reads from the current position to the end (of the file / of the range); checks the records against want[at ...]; returns how many it saw
            auto pass = [&](int mode, int64_t batch, size_t at, long long stop_after) -> long long {
                long long got = 0; svx_batch prev; bool have_prev = false; unsigned long long prev_sum = 0; std::string last_name;
                for (long long k = 0; stop_after < 0 || k < stop_after; k++) {
                    svx_batch b; int64_t n = 0;
                    if (svx_bam_read_batch(h, batch, mode, 20, &b, &n) != SVX_OK) { dbad++; fprintf(stderr, "device mode, chunks of %s blocks: %s\n", chunk_blocks[cb], svx_last_error()); return -1; }
                    if (have_prev && touch(prev) != prev_sum) { dbad++; fprintf(stderr, "chunks of %s blocks: the batch handed out before changed\n", chunk_blocks[cb]); }
                    if (n == 0) break;
                    const std::vector<std::string> nm = names_now();
                    for (int64_t i = 0; i < n; i++) {
                        const size_t w = at + (size_t)got + (size_t)i;
                        if (w >= want.size() || b.tid[i] != want[w].tid || b.pos[i] != want[w].pos || b.lseq[i] != want[w].lseq || (b.flag[i] & 0x0fff) != want[w].flag ||
                            b.cigar_off[i + 1] - b.cigar_off[i] != want[w].ncig || nm[(size_t)b.read_id[i]] != want[w].name) { dbad++; fprintf(stderr, "chunks of %s blocks, mode %d: record %zu differs\n", chunk_blocks[cb], mode, w); return -1; }
                    }
                    if (mode == 1 && !last_name.empty() && nm[(size_t)b.read_id[0]] == last_name) { dbad++; fprintf(stderr, "chunks of %s blocks: a read group was split between batches\n", chunk_blocks[cb]); }
                    last_name = nm[(size_t)b.read_id[n - 1]];
                    got += n; prev = b; have_prev = true; prev_sum = touch(b);
                }
                return got;
            };
            if (pass(0, 37, 0, -1) != (long long)want.size()) { dbad++; fprintf(stderr, "chunks of %s blocks: coordinate pass incomplete\n", chunk_blocks[cb]); }
            if (svx_bam_rewind(h) != SVX_OK) dbad++;
            (void)pass(0, 11, 0, 5);                                  // leave in the middle of a pass (a prefetch in flight), then start again in the other mode
            if (svx_bam_rewind(h) != SVX_OK) dbad++;
            if (pass(1, 29, 0, -1) != (long long)want.size()) { dbad++; fprintf(stderr, "chunks of %s blocks: query-name pass incomplete\n", chunk_blocks[cb]); }
            for (int round = 0; round < 2; round++)
                for (size_t k = 0; k < tids.size(); k++) {
                    const size_t r = (k * 7 + 3 + (size_t)round) % tids.size();
                    if (svx_bam_seek(h, voff_of(first_of[r]), tids[r]) != SVX_OK) { dbad++; continue; }
                    if (pass(0, 23, first_idx[r], -1) != count_of[r]) { dbad++; fprintf(stderr, "chunks of %s blocks: range of tid %d incomplete\n", chunk_blocks[cb], tids[r]); }
                }
            svx_bam_close(h);
        }
        unsetenv("SVX_BAM_DEV_CHUNK_BLOCKS");
        printf("device reader's host side over the stand-in decoder: %lld problems\n", dbad);
        return bad || rbad || dbad ? 1 : 0;
    }
    long long ok = 0, failed = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> img;
        if (it & 1) {                                                  // the compressed file itself
            img = file;
            const int k = 1 + (int)(rnd() % 6);
            for (int j = 0; j < k; j++) img[rnd() % img.size()] = (uint8_t)rnd();
            if (rnd() % 5 == 0) img.resize(rnd() % img.size());
        } else {                                                       // the record stream
            std::vector<uint8_t> r = raw;
            const int k = 1 + (int)(rnd() % 4);
            for (int j = 0; j < k && !rec.empty(); j++) {
                const size_t p = rec[rnd() % rec.size()];
                const uint32_t extreme[6] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0xffffffffu, (uint32_t)rnd()};
                switch (rnd() % 7) {
                    case 0: { const uint32_t v = extreme[rnd() % 6]; memcpy(r.data() + p, &v, 4); break; }                       // block_size
                    case 1: { const uint32_t v = extreme[rnd() % 6]; memcpy(r.data() + p + 4 + 4 * (rnd() % 8), &v, 4); break; }   // a fixed field (refID .. tlen)
                    case 2: r[p + 12] = (uint8_t)rnd(); break;                                                                    // l_read_name
                    case 3: { const uint16_t v = (uint16_t)rnd(); memcpy(r.data() + p + 16, &v, 2); break; }                      // n_cigar_op
                    case 4: { const size_t q = p + 36 + rnd() % 600; if (q < r.size()) r[q] = (uint8_t)rnd(); break; }            // name / CIGAR / SEQ / tags
                    case 5: { const size_t q = p + 36 + rnd() % 4000; for (size_t t = q; t < r.size() && t < q + 32; t++) r[t] = (uint8_t)rnd(); break; }
                    default: { for (int t = 0; t < 16; t++) { const size_t q = rnd() % r.size(); r[q] = (uint8_t)rnd(); } break; }
                }
            }
            if (rnd() % 6 == 0) r.resize(8 + rnd() % (r.size() - 8));
            img = bgzf_all(r, 2000 + rnd() % 60000);
        }
        FILE* f = fopen(tmp.c_str(), "wb"); fwrite(img.data(), 1, img.size(), f); fclose(f);
        const long long got = read_to_end(tmp.c_str(), (int)(rnd() & 1), (int)(rnd() % 3 == 0), 1 + (int)(rnd() % 3));
        if (got >= 0) ok++; else failed++;
    }
    remove(tmp.c_str());
    printf("damaged files: %d, read to the end %lld, refused %lld\n", iters, ok, failed);
    return 0;
}
