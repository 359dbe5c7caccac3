// Host build of svim_amd/csrc/inflate_lanes.hpp (the lane-per-block DEFLATE decoder of the GPU, every lane a serial decoder), checked against zlib:
//   inflate_lanes_host_test <file.bam>     every BGZF block of the file: decoded == zlib, or given up (counted; the GPU redoes those with the wave-per-block decoder)
//   inflate_lanes_host_test --fuzz N       N random buffers deflated at levels 0..9 / every strategy
//   inflate_lanes_host_test --damaged N    damaged streams: an answer or a refusal, never a write outside the output and never more than a bounded number of trips
// Build: g++ -O2 -std=c++17 -DINFL_HOST -I svim_amd/csrc tools/inflate_lanes_host_test.cpp -lz -o /tmp/inflate_lanes_host_test
#include <cstdint>
static unsigned long long g_sub_hist[2][9], g_stall, g_trips, g_hdrs, g_sub_max[2];
static inline void stat_(int what, unsigned n) {
    if (what < 2) { int b = 0; while ((32u << b) < n && b < 8) b++; g_sub_hist[what][n ? b : 0]++; if (n > g_sub_max[what]) g_sub_max[what] = n; }
    else g_stall += n;
}
#define INFL_STAT(what, n) stat_(what, n)
static unsigned long long g_fail_site[16];
#define INFL_FAIL(L, site) do { (L).state = 3u; g_fail_site[site]++; } while (0)
#define INFL_HOST 1
#include "inflate_lanes.hpp"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
static_assert(INFL_DEPTH <= 6, "the switch below");

// 1 = decoded and identical, 0 = given up (block goes to the other decoder), -1 = WRONG (a result that differs, or a write outside the output)
static int run_lane(const uint8_t* comp, size_t clen, const std::vector<uint8_t>* expect, size_t cap, unsigned shift) {
    std::vector<uint64_t> in((clen + 7) / 8 + 6, 0);
    uint8_t* payload = reinterpret_cast<uint8_t*>(in.data()) + (shift & 7u);
    memcpy(payload, comp, clen);
    std::vector<uint8_t> out(cap + 16, 0xAA);
    std::vector<uint32_t> scratch(INFL_BYTES / 4 + 1, 0xDEADBEEF);
    std::vector<uint32_t> gscratch(3 * INFL_LENS_WORDS + 1, 0xDEADBEEF);        // the lane's global scratch, at a stride of 3 words
    InflLane L;
    infl_init(L, payload, (uint32_t)clen, out.data(), (uint32_t)cap, reinterpret_cast<uint8_t*>(scratch.data()), gscratch.data(), 3u);
    unsigned long long trips = 0;
    const unsigned long long bound = 4ull * cap + 4096ull;
    while (infl_running(L)) {
        if (L.state == INFL_ST_HEADER) { infl_header(L); g_hdrs++; }
        switch (trips % INFL_DEPTH) {                                        // (the GPU unrolls its loop: the slot of a trip is a constant)
            case 0: infl_step<0>(L); break;
            case 1: infl_step<1 % INFL_DEPTH>(L); break;
            case 2: infl_step<2 % INFL_DEPTH>(L); break;
            case 3: infl_step<3 % INFL_DEPTH>(L); break;
            case 4: infl_step<4 % INFL_DEPTH>(L); break;
            default: infl_step<5 % INFL_DEPTH>(L); break;
        }
        if (++trips > bound) { fprintf(stderr, "trip bound exceeded\n"); return -1; }
    }
    g_trips += trips;
    if (scratch[INFL_BYTES / 4] != 0xDEADBEEF || gscratch[3 * INFL_LENS_WORDS] != 0xDEADBEEF) { fprintf(stderr, "scratch overrun\n"); return -1; }
    for (size_t i = cap; i < out.size(); i++) if (out[i] != 0xAA) { fprintf(stderr, "wrote behind the output\n"); return -1; }
    if (L.state != INFL_ST_DONE) return 0;
    if (expect && (L.pos != expect->size() || memcmp(out.data(), expect->data(), expect->size()) != 0)) {
        size_t d = 0; while (d < expect->size() && out[d] == (*expect)[d]) d++;
        fprintf(stderr, "MISMATCH: %u bytes for %zu, first difference at %zu\n", L.pos, expect->size(), d);
        return -1;
    }
    return 1;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& src, int level, int strategy) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(2 * src.size() + 4096);
    zs.next_in = const_cast<Bytef*>(src.data()); zs.avail_in = (uInt)src.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "deflate did not finish\n"); exit(2); }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}
static bool zlib_ok(const std::vector<uint8_t>& comp, size_t cap, std::vector<uint8_t>& out) {
    out.assign(cap + 1, 0);
    z_stream zs; memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, -15);
    zs.next_in = const_cast<Bytef*>(comp.data()); zs.avail_in = (uInt)comp.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == cap;
    out.resize(zs.total_out);
    inflateEnd(&zs);
    return ok;
}

static unsigned long long g_x = 88172645463325252ull;
static unsigned long long rnd() { g_x ^= g_x << 13; g_x ^= g_x >> 7; g_x ^= g_x << 17; return g_x; }
static void fill(std::vector<uint8_t>& src, int kind) {
    const size_t len = src.size();
    for (size_t i = 0; i < len; i++) {
        switch (kind) {
            case 0: src[i] = (uint8_t)rnd(); break;
            case 1: src[i] = (uint8_t)("ACGT"[rnd() & 3]); break;
            case 2: src[i] = (uint8_t)(i % 7 == 0 ? rnd() : 'I'); break;
            case 3: src[i] = (uint8_t)(i >= 300 && (rnd() % 5) ? src[i - 1 - rnd() % 299] : rnd()); break;
            case 4: src[i] = (uint8_t)(i & 1 ? 0 : rnd() % 3); break;
            case 5: src[i] = (uint8_t)(33 + rnd() % 40); break;
            case 6: { const unsigned q = 2 + (unsigned)((rnd() % 12) + (rnd() % 12) + (rnd() % 12));
                      src[i] = (uint8_t)(i >= 40 && rnd() % 9 == 0 ? src[i - 3 - rnd() % 37] : q); break; }
            case 7: src[i] = (uint8_t)(i >= 3 && rnd() % 4 ? src[i - 1 - rnd() % 3] : rnd()); break;                  // distances 1..3: the pattern rules of the copy engine
            default: src[i] = (uint8_t)((i / 700) & 1 ? 1 + rnd() % 45 : (rnd() % 11 == 0 ? rnd() : "\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88"[rnd() & 15])); break;
        }
    }
}

int main(int argc, char** argv) {
    const int strategies[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
    if (argc >= 3 && std::string(argv[1]) == "--fuzz") {
        const int n = atoi(argv[2]);
        int bad = 0, gave_up = 0, ok = 0;
        for (int it = 0; it < n; it++) {
            std::vector<uint8_t> src((size_t)(rnd() % 65281));
            fill(src, (int)(rnd() % 9));
            const int level = (int)(rnd() % 10);
            const std::vector<uint8_t> comp = deflate_raw(src, level, strategies[rnd() % 4]);
            const int rc = run_lane(comp.data(), comp.size(), &src, src.size(), (unsigned)it);
            if (rc < 0) { bad++; fprintf(stderr, "  (buffer %d, %zu bytes, level %d)\n", it, src.size(), level); }
            else if (rc == 0) gave_up++; else ok++;
        }
        printf("fuzz: %d buffers, %d mismatches, %d decoded, %d given up (stored blocks, codes beyond the table budget)\n", n, bad, ok, gave_up);
        return bad ? 1 : 0;
    }
    if (argc >= 3 && std::string(argv[1]) == "--damaged") {
        const int n = atoi(argv[2]);
        int bad = 0, done = 0, answered = 0, agree = 0;
        for (int it = 0; it < n; it++) {
            std::vector<uint8_t> src((size_t)(1 + rnd() % 40000));
            fill(src, (int)(rnd() % 9));
            const std::vector<uint8_t> good = deflate_raw(src, (int)(rnd() % 10), strategies[rnd() % 4]);
            for (int v = 0; v < 6; v++) {
                std::vector<uint8_t> c = good;
                size_t cap = src.size();
                switch (v) {
                    case 0: for (int k = 0; k < 1 + (int)(rnd() % 4); k++) c[rnd() % c.size()] ^= (uint8_t)(1u << (rnd() & 7)); break;
                    case 1: for (int k = 0; k < 8 && k < (int)c.size(); k++) c[k] = (uint8_t)rnd(); break;
                    case 2: c.resize(rnd() % c.size()); break;
                    case 3: cap = rnd() % (src.size() + 1); break;
                    case 4: { const size_t at = rnd() % c.size(); for (size_t k = at; k < c.size() && k < at + 64; k++) c[k] = (uint8_t)rnd(); break; }
                    default: for (size_t k = 0; k < c.size(); k++) if (rnd() % 97 == 0) c[k] = (uint8_t)rnd(); break;
                }
                // a lane may only ANSWER (state DONE) what zlib also accepts with the same bytes; anything else must be a refusal
                std::vector<uint8_t> z;
                const bool zok = zlib_ok(c, cap, z);
                const int rc = run_lane(c.data(), c.size(), zok ? &z : nullptr, cap, (unsigned)it);
                done++;
                if (rc < 0) bad++;
                else if (rc == 1) { answered++; if (zok) agree++; else { bad++; fprintf(stderr, "answered a stream zlib refuses\n"); } }
            }
        }
        printf("damaged: %d streams, %d bad, %d answered (%d of them accepted by zlib with the same bytes)\n", done, bad, answered, agree);
        return bad ? 1 : 0;
    }
    if (argc < 2) { fprintf(stderr, "usage: %s <file.bam> | --fuzz N | --damaged N\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    std::vector<uint8_t> file;
    { uint8_t buf[1 << 16]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + k); }
    fclose(f);
    size_t at = 0; int blocks = 0, bad = 0, gave_up = 0; unsigned long long out_bytes = 0, in_bytes = 0;
    const int max_blocks = argc >= 3 ? atoi(argv[2]) : 1 << 30;
    while (at + 18 <= file.size() && blocks < max_blocks) {
        const uint8_t* h = file.data() + at;
        if (h[0] != 0x1f || h[1] != 0x8b) { fprintf(stderr, "not a BGZF block at %zu\n", at); return 2; }
        const unsigned xlen = h[10] | (h[11] << 8);
        unsigned bsize = 0;
        for (unsigned p = 12; p + 4 <= 12 + xlen;) { const unsigned sl = h[p + 2] | (h[p + 3] << 8); if (h[p] == 'B' && h[p + 1] == 'C') bsize = (h[p + 4] | (h[p + 5] << 8)) + 1u; p += 4 + sl; }
        if (!bsize) { fprintf(stderr, "no BC field\n"); return 2; }
        const uint8_t* payload = h + 12 + xlen; const size_t clen = bsize - 12 - xlen - 8;
        const uint8_t* tr = h + bsize - 8;
        const uint32_t isize = tr[4] | (tr[5] << 8) | (tr[6] << 16) | ((uint32_t)tr[7] << 24);
        std::vector<uint8_t> expect(isize + 1u);
        { z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15); zs.next_in = const_cast<Bytef*>(payload); zs.avail_in = (uInt)clen; zs.next_out = expect.data(); zs.avail_out = isize + 1u;
          const int rc = inflate(&zs, Z_FINISH); const bool ok = rc == Z_STREAM_END && zs.total_out == isize; inflateEnd(&zs); if (!ok) { fprintf(stderr, "zlib refuses block %d\n", blocks); return 2; } }
        expect.resize(isize);
        const int rc = run_lane(payload, clen, &expect, isize, (unsigned)blocks);
        if (rc < 0) bad++; else if (rc == 0) gave_up++;
        blocks++; out_bytes += isize; in_bytes += clen; at += bsize;
    }
    printf("%d blocks, %d mismatches, %d given up; %.1f MB -> %.1f MB\n", blocks, bad, gave_up, in_bytes / 1e6, out_bytes / 1e6);
    printf("trips per block %.0f (%.2f output bytes per trip), trips a length code waited for the copy engine %.1f %%, block headers per BGZF block %.2f\n",
           (double)g_trips / blocks, (double)out_bytes / g_trips, 100.0 * g_stall / g_trips, (double)g_hdrs / blocks);
    printf("given up by site (1-6 header fields, 7 stored block, 8 table budget / over-subscribed, 9-13 stream):");
    for (int k = 1; k < 14; k++) printf(" %llu", g_fail_site[k]);
    printf("\n");
    for (int t = 0; t < 2; t++) {
        printf("%s second-level entries used (0, <=32, <=64, <=128, <=256, ...):", t ? "distance" : "literal/length");
        for (int b = 0; b < 9; b++) printf(" %llu", g_sub_hist[t][b]);
        printf("   max %llu\n", g_sub_max[t]);
    }
    return bad ? 1 : 0;
}
