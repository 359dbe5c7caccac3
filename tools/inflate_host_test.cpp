// Host build of svim_amd/csrc/inflate_core.hpp (the GPU's DEFLATE decoder with the lane operations emulated), checked against zlib:
//   inflate_host_test <file.bam|file.gz-with-BGZF-blocks>   every BGZF block of the file: inflate_raw == zlib
//   inflate_host_test --fuzz N                              N random buffers deflated at levels 0..9 / strategies, incl. stored and fixed blocks
// Build: g++ -O2 -std=c++17 -DINF_HOST -I svim_amd/csrc tools/inflate_host_test.cpp -lz -o /tmp/inflate_host_test
#include <cstdint>
static unsigned long long g_tok[2][3], g_len[2][10], g_dist[2][17], g_bytes[2];      // [path: 0 token chain, 1 serial][...]
static inline void stat_token(int path, unsigned len, unsigned dist) {
    g_tok[path][len ? 1 : 0]++;
    g_bytes[path] += len ? len : 1;
    if (!len) return;
    int lb = 0; while ((1u << (lb + 1)) <= len && lb < 9) lb++;
    int db = 0; while ((1u << (db + 1)) <= dist && db < 16) db++;
    g_len[path][lb]++; g_dist[path][db]++;
}
#define INF_STAT(path, len, dist) stat_token(path, len, dist)
static unsigned long long g_wide[8];                                                   // multi-window steps: [given up, -, taken with 2, 3, 4 windows]
#define INF_WIDE_STAT(taken) g_wide[taken]++;
static unsigned long long g_steps;
#define INF_STEP_STAT() g_steps++;
#include "inflate_core.hpp"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static int check(const uint8_t* comp, size_t clen, const std::vector<uint8_t>& expect, const char* what) {
    static unsigned shift = 0;                                          // every alignment of the payload gets its turn
    std::vector<uint32_t> in((clen + 3) / 4 + 4, 0);
    uint8_t* payload = reinterpret_cast<uint8_t*>(in.data()) + (shift++ & 3u);
    memcpy(payload, comp, clen);
    std::vector<uint8_t> out(expect.size() + 8, 0xAA);
    InfScratch sc;
    const int rc = inflate_raw(payload, (uint32_t)clen, out.data(), (uint32_t)expect.size(), sc);
    if (rc != (int)expect.size() || memcmp(out.data(), expect.data(), expect.size()) != 0 || out[expect.size()] != 0xAA) {
        size_t d = 0; while (d < expect.size() && out[d] == expect[d]) d++;
        fprintf(stderr, "MISMATCH %s: rc %d expected %zu, first difference at %zu\n", what, rc, expect.size(), d);
        return 1;
    }
    return 0;
}

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& src, int level, int strategy) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
    std::vector<uint8_t> out(2 * src.size() + 4096);                      // fixed codes can expand incompressible data beyond deflateBound
    zs.next_in = const_cast<Bytef*>(src.data()); zs.avail_in = (uInt)src.size(); zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "deflate did not finish\n"); exit(2); }
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

// a damaged stream must end in an error code or in some output - never in an access outside its buffers (build with -fsanitize=address,undefined) and
// never in more than the output capacity written; returns 1 if the guard bytes behind the output were touched
static int check_damaged(const std::vector<uint8_t>& comp, size_t out_cap) {
    std::vector<uint32_t> in((comp.size() + 3) / 4 + 1, 0);
    memcpy(in.data(), comp.data(), comp.size());
    std::vector<uint8_t> out(out_cap + 8, 0xAA);
    InfScratch sc;
    (void)inflate_raw(reinterpret_cast<uint8_t*>(in.data()), (uint32_t)comp.size(), out.data(), (uint32_t)out_cap, sc);
    for (size_t i = out_cap; i < out.size(); i++) if (out[i] != 0xAA) return 1;
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && std::string(argv[1]) == "--damaged") {
        // N valid streams, each damaged a few times: bit flips, overwritten bytes, truncation, a lying output capacity
        const int n = atoi(argv[2]);
        unsigned long long x = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        int bad = 0, done = 0;
        for (int it = 0; it < n; it++) {
            const size_t len = (size_t)(1 + rnd() % 40000);
            std::vector<uint8_t> src(len);
            const int kind = (int)(rnd() % 4);
            for (size_t i = 0; i < len; i++)
                src[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? (uint8_t)("ACGT"[rnd() & 3]) : kind == 2 ? (uint8_t)(i >= 300 && (rnd() % 5) ? src[i - 1 - rnd() % 299] : rnd())
                                                                                                          : (uint8_t)(33 + rnd() % 40);
            const int strategies[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
            const std::vector<uint8_t> good = deflate_raw(src, (int)(rnd() % 10), strategies[rnd() % 4]);
            for (int v = 0; v < 6; v++) {
                std::vector<uint8_t> c = good;
                size_t cap = len;
                switch (v) {
                    case 0: for (int k = 0; k < 1 + (int)(rnd() % 4); k++) c[rnd() % c.size()] ^= (uint8_t)(1u << (rnd() & 7)); break;          // bit flips anywhere
                    case 1: for (int k = 0; k < 8 && k < (int)c.size(); k++) c[k] = (uint8_t)rnd(); break;                                   // a damaged block header
                    case 2: c.resize(rnd() % c.size()); break;                                                                              // cut short
                    case 3: cap = rnd() % (len + 1); break;                                                                                 // the ISIZE field lies (too small)
                    case 4: { const size_t at = rnd() % c.size(); for (size_t k = at; k < c.size() && k < at + 64; k++) c[k] = (uint8_t)rnd(); break; }
                    default: for (size_t k = 0; k < c.size(); k++) if (rnd() % 97 == 0) c[k] = (uint8_t)rnd(); break;
                }
                bad += check_damaged(c, cap); done++;
            }
        }
        printf("damaged: %d streams, %d wrote behind their output\n", done, bad);
        return bad ? 1 : 0;
    }
    if (argc >= 3 && std::string(argv[1]) == "--fuzz") {
        const int n = atoi(argv[2]);
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        int bad = 0, done = 0;
        for (int it = 0; it < n; it++) {
            const size_t len = (size_t)(rnd() % 65281);
            std::vector<uint8_t> src(len);
            const int kind = (int)(rnd() % 8);
            for (size_t i = 0; i < len; i++) {
                switch (kind) {
                    case 0: src[i] = (uint8_t)rnd(); break;                                       // incompressible
                    case 1: src[i] = (uint8_t)("ACGT"[rnd() & 3]); break;
                    case 2: src[i] = (uint8_t)(i % 7 == 0 ? rnd() : 'I'); break;                  // long runs
                    case 3: src[i] = (uint8_t)(i >= 300 && (rnd() % 5) ? src[i - 1 - rnd() % 299] : rnd()); break;   // many short matches
                    case 4: src[i] = (uint8_t)(i & 1 ? 0 : rnd() % 3); break;
                    case 5: src[i] = (uint8_t)(33 + rnd() % 40); break;                           // quality-like
                    case 6: { const unsigned q = 2 + (unsigned)((rnd() % 12) + (rnd() % 12) + (rnd() % 12));                  // Phred values with a bell-shaped law: literals
                              src[i] = (uint8_t)(i >= 40 && rnd() % 9 == 0 ? src[i - 3 - rnd() % 37] : q); break; }          //   with short matches in between (multi-window steps)
                    default: src[i] = (uint8_t)((i / 700) & 1 ? 1 + rnd() % 45 : (rnd() % 11 == 0 ? rnd() : "\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88"[rnd() & 15])); break;   // BAM-like: packed bases, then qualities
                }
            }
            const int level = (int)(rnd() % 10);
            const int strategies[4] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
            const int strat = strategies[rnd() % 4];
            const std::vector<uint8_t> comp = deflate_raw(src, level, strat);
            char what[96]; snprintf(what, sizeof what, "fuzz %d kind %d level %d strategy %d len %zu", it, kind, level, strat, len);
            bad += check(comp.data(), comp.size(), src, what); done++;
        }
        printf("fuzz: %d buffers, %d mismatches\n", done, bad);
        printf("multi-window steps: %llu taken (2 / 3 / 4 windows: %llu / %llu / %llu), %llu given up\n", g_wide[2] + g_wide[3] + g_wide[4], g_wide[2], g_wide[3], g_wide[4], g_wide[0]);
        return bad ? 1 : 0;
    }
    if (argc < 2) { fprintf(stderr, "usage\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> file;
    { uint8_t buf[1 << 16]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + k); }
    fclose(f);
    size_t at = 0; int blocks = 0, bad = 0; size_t total = 0;
    while (at + 18 <= file.size()) {
        const uint8_t* hd = file.data() + at;
        if (hd[0] != 31 || hd[1] != 139) { fprintf(stderr, "not a BGZF block at %zu\n", at); return 2; }
        const unsigned xlen = hd[10] | (hd[11] << 8);
        int bsize = -1;
        for (size_t p = 0; p + 4 <= xlen;) { const uint8_t* e = hd + 12 + p; const unsigned sl = e[2] | (e[3] << 8); if (e[0] == 'B' && e[1] == 'C') bsize = e[4] | (e[5] << 8); p += 4 + sl; }
        if (bsize < 0) { fprintf(stderr, "no BC field\n"); return 2; }
        const size_t blen = (size_t)bsize + 1, cstart = 12 + xlen, clen = blen - cstart - 8;
        const uint32_t isize = hd[blen - 4] | (hd[blen - 3] << 8) | (hd[blen - 2] << 16) | ((uint32_t)hd[blen - 1] << 24);
        std::vector<uint8_t> expect(isize);
        z_stream zs; memset(&zs, 0, sizeof zs);
        inflateInit2(&zs, -15);
        zs.next_in = const_cast<Bytef*>(hd + cstart); zs.avail_in = (uInt)clen; zs.next_out = expect.data(); zs.avail_out = isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END && isize) { fprintf(stderr, "zlib failed on block %d\n", blocks); return 2; }
        char what[64]; snprintf(what, sizeof what, "block %d", blocks);
        bad += check(hd + cstart, clen, expect, what);
        at += blen; blocks++; total += isize;
    }
    printf("%s: %d BGZF blocks, %zu bytes inflated, %d mismatches\n", argv[1], blocks, total, bad);
    printf("decode steps: %llu (%.1f per block, %.1f bytes per step)\n", g_steps, (double)g_steps / (blocks ? blocks : 1), (double)total / (g_steps ? g_steps : 1));
    printf("multi-window steps: %llu taken (2 / 3 / 4 windows: %llu / %llu / %llu), %llu given up\n", g_wide[2] + g_wide[3] + g_wide[4], g_wide[2], g_wide[3], g_wide[4], g_wide[0]);
    for (int p = 0; p < 2; p++) {
        printf("%s: %llu literals, %llu matches, %llu bytes\n  match length (log2 bins from 2):", p ? "serial tokens" : "token chain", g_tok[p][0], g_tok[p][1], g_bytes[p]);
        for (int i = 1; i < 10; i++) printf(" %llu", g_len[p][i]);
        printf("\n  distance (log2 bins from 1):");
        for (int i = 0; i < 17; i++) printf(" %llu", g_dist[p][i]);
        printf("\n");
    }
    return bad ? 1 : 0;
}
