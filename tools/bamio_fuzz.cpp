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
extern "C" int svx_inflater_pin(svx_inflater*, void*, uint64_t) { return SVX_E_NODEVICE; }
extern "C" int svx_inflater_unpin(svx_inflater*, void*) { return SVX_E_NODEVICE; }
extern "C" int svx_inflater_enqueue(svx_inflater*, int, int64_t, const uint64_t*, const uint32_t*, const uint32_t*, const uint64_t*, uint64_t, uint8_t*, uint64_t, int) { return SVX_E_NODEVICE; }
extern "C" int svx_inflater_wait(svx_inflater*, int, float*) { return SVX_E_NODEVICE; }
extern "C" long long svx_inflater_unregister_failures() { return 0; }
int devdec_create(int, int, int32_t, const int32_t*, const char*, const int32_t*, svx_devdec**) { return SVX_E_NODEVICE; }
void devdec_destroy(svx_devdec*) {}
void devdec_set_file(svx_devdec*, const uint8_t*, size_t) {}
int devdec_load(svx_devdec*, int, const DevDecBlock*, size_t, int, uint64_t, bool, int, int) { return SVX_E_NODEVICE; }
int devdec_count(svx_devdec*, int, int32_t, int64_t*, int64_t*) { return SVX_E_NODEVICE; }
int devdec_batch(svx_devdec*, int, int64_t, int64_t*, int, int, svx_batch*) { return SVX_E_NODEVICE; }
const std::vector<std::string>& devdec_names(svx_devdec*) { static std::vector<std::string> none; return none; }
void devdec_stats(svx_devdec*, DevDecStats*) {}
void devdec_reset_names(svx_devdec*) {}

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
        return bad || rbad ? 1 : 0;
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
