/* stair_model.c - a cell-level model of the narrowing staircase window of svim_amd/csrc/edit.hip (d_edit_stair / stair_block), checked against the plain
 * dynamic programme: whenever the model ACCEPTS a result (d <= kcap, the kernel's rule) it must be the edit distance, for every upper bound ub >= distance the
 * pair may come with.  The kernel computes the same recurrence bit-parallel (Myers / Hyyro); what is modelled here is everything else - the geometry of the
 * window, its boundary assumptions (the row above steps +1 per column, an entering word +1 per row, virtual rows <= 0 and > m never match) and the two cut-off
 * rules that make it narrow - so that the RULES are tested on millions of cells independently of the GPU (tests/test_host_cpu.py runs it).
 *   cc -O2 tools/stair_model.c -o stair_model && ./stair_model <cases> [seed]                                                                              */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define QMIN 3
#define MAXQ 16
#define BIG 1000000
static unsigned long long rs = 88172645463325252ull;
static unsigned long long rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }

static int plain(const char* p, int m, const char* t, int n) {
    static int a[8192], b[8192];
    int* prev = a; int* cur = b;
    for (int j = 0; j <= n; j++) prev[j] = j;
    for (int i = 1; i <= m; i++) {
        cur[0] = i;
        for (int j = 1; j <= n; j++) { int v = prev[j - 1] + (p[i - 1] != t[j - 1]); if (prev[j] + 1 < v) v = prev[j] + 1; if (cur[j - 1] + 1 < v) v = cur[j - 1] + 1; cur[j] = v; }
        int* s = prev; prev = cur; cur = s;
    }
    return prev[n];
}

/* returns the accepted distance, or -1 (the kernel would send the pair to a wider class); *cells = window cells computed */
static int stair(const char* p, int m, const char* t, int n, int Q0, int ub, int narrow, long long* cells) {
    const int W0 = 32 * Q0, delta = n - m;
    int off = (W0 - 34 + delta) / 2;
    if (off < 7) off = 7;
    off -= ((off - 7) & 7);
    const int margin_lo = W0 - off - 33, margin_up = off + 1 - delta;
    const int margin = margin_lo < margin_up ? margin_lo : margin_up;
    int kcap = margin >= 0 ? delta + 2 * margin + 1 : -1;
    if (ub < kcap) kcap = ub;
    static int col[32 * MAXQ], nxt[32 * MAXQ];
    int Q = Q0, trow = -off, top = off + 1, d = -1;
    for (int r = 0; r < 32 * Q; r++) { const int i = trow + r; col[r] = i < 0 ? -i : i; }       /* column 0: |i| (virtual rows above row 0 step -1) */
    const int n_blocks = (n + 31) >> 5;
    for (int kb = 0; kb < n_blocks; kb++) {
        const int W = 32 * Q;
        for (int j = 32 * kb + 1; j <= 32 * kb + 32 && j <= n; j++) {
            const int top_prev = top; top += 1;
            for (int r = 0; r < W; r++) {
                const int i = trow + r;
                const int up = r ? nxt[r - 1] : top, diag = r ? col[r - 1] : top_prev, left = col[r];
                const int hit = i >= 1 && i <= m && p[i - 1] == t[j - 1];
                int v = diag + !hit; if (up + 1 < v) v = up + 1; if (left + 1 < v) v = left + 1;
                nxt[r] = v;
            }
            memcpy(col, nxt, sizeof(int) * (size_t)W);
            *cells += W;
            if (j == n) { const int bm = m - trow; d = (bm >= 0 && bm < W) ? col[bm] : -1; }
        }
        const int j1 = 32 * (kb + 1);
        if (j1 >= n) break;
        const int t1 = col[31];
        int et = 0, eb = 0;
        if (narrow && Q > QMIN) {
            const int t2 = col[63], oc_t = j1 - (trow + 63);
            et = oc_t >= delta && t2 + (oc_t - delta) > kcap;
            const int sb = col[32 * (Q - 1) - 1], bot = trow + 32 * Q, oc_b = j1 - (bot - 33);
            eb = bot > m || (oc_b <= delta && sb + (delta - oc_b) > kcap);
            if (et && eb && Q - 2 < QMIN) eb = 0;
        }
        const int drop = et ? 64 : 32;
        top = et ? col[63] : t1;
        const int keep = W - drop;
        memmove(col, col + drop, sizeof(int) * (size_t)keep);
        trow += drop;
        int Wn = keep;
        if (!eb) { for (int r = 0; r < 32; r++) col[keep + r] = col[keep - 1] + 1 + r; Wn += 32; }        /* an entering word steps +1 per row */
        Q = Wn / 32;
    }
    if (margin >= 0 && d >= 0 && d <= kcap) return d;
    return -1;
}

static void mutate(char* s, int* len, int cap, double rate, int lo, int hi) {
    int k = (int)(rate * (hi - lo));
    for (int e = 0; e < k; e++) {
        if (*len <= 1) break;
        int p = lo + (int)(rnd() % (unsigned)((hi < *len ? hi : *len) - lo > 0 ? (hi < *len ? hi : *len) - lo : 1));
        if (p >= *len) p = *len - 1;
        const unsigned r = (unsigned)(rnd() % 10);
        if (r < 4) s[p] = "ACGT"[rnd() & 3];
        else if (r < 7) { memmove(s + p, s + p + 1, (size_t)(*len - p - 1)); (*len)--; }
        else if (*len + 1 < cap) { memmove(s + p + 1, s + p, (size_t)(*len - p)); s[p] = "ACGT"[rnd() & 3]; (*len)++; }
    }
}

int main(int argc, char** argv) {
    const long cases = argc > 1 ? atol(argv[1]) : 20000;
    if (argc > 2) rs ^= (unsigned long long)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    long accepted = 0, refused = 0, wrong = 0, narrowed_fewer = 0;
    long long cells_static = 0, cells_narrow = 0;
    static char a[4096], b[4096];
    for (long it = 0; it < cases; it++) {
        int la = 40 + (int)(rnd() % 900);
        const int kind = (int)(rnd() % 6);
        for (int i = 0; i < la; i++) a[i] = kind == 5 ? "AC"[(i / (1 + (int)(it % 7))) & 1] : "ACGT"[rnd() & 3];          /* kind 5: low complexity (many equally good paths) */
        int lb = la; memcpy(b, a, (size_t)la);
        switch (kind) {
            case 0: mutate(b, &lb, 4000, 0.01 * (double)(rnd() % 12), 0, lb); break;                                        /* noise everywhere */
            case 1: mutate(b, &lb, 4000, 0.3, 0, lb / 6 + 1); break;                                                        /* bunched at the start */
            case 2: mutate(b, &lb, 4000, 0.3, lb - lb / 6 - 1, lb); break;                                                  /* ... at the end */
            case 3: { int g = 5 + (int)(rnd() % 60); const int at = (int)(rnd() % (unsigned)(lb / 3 + 1));                  /* a deletion early, paid back by an insertion late */
                      if (g > lb - at - 10) g = lb - at - 10 > 0 ? lb - at - 10 : 0;
                      memmove(b + at, b + at + g, (size_t)(lb - at - g)); lb -= g;
                      const int at2 = lb - (int)(rnd() % (unsigned)(lb / 3 + 1)); memmove(b + at2 + g, b + at2, (size_t)(lb - at2)); for (int i = 0; i < g; i++) b[at2 + i] = "ACGT"[rnd() & 3]; lb += g;
                      mutate(b, &lb, 4000, 0.02, 0, lb); break; }
            case 4: { const int sh = 1 + (int)(rnd() % 40); memmove(b + sh, b, (size_t)lb); for (int i = 0; i < sh; i++) b[i] = "ACGT"[rnd() & 3]; lb += sh; mutate(b, &lb, 4000, 0.03, 0, lb); break; }
            default: mutate(b, &lb, 4000, 0.05, 0, lb); break;
        }
        if (rnd() % 4 == 0) { const int cut = (int)(rnd() % 120); if (lb - cut > 20) lb -= cut; }                              /* unequal lengths */
        const char* p = la <= lb ? a : b; const char* t = la <= lb ? b : a;
        const int m = la <= lb ? la : lb, n = la <= lb ? lb : la;
        const int truth = plain(p, m, t, n);
        for (int v = 0; v < 4; v++) {
            const int Q0 = 4 + 2 * (int)(rnd() % 4);                                                                        /* 4, 6, 8, 10 words */
            const int ub = v == 0 ? truth : v == 1 ? truth + (int)(rnd() % 30) : v == 2 ? n : truth + (int)(rnd() % 4);       /* any valid upper bound: exact, loose, useless, nearly exact */
            long long c0 = 0, c1 = 0;
            const int s0 = stair(p, m, t, n, Q0, ub, 0, &c0), s1 = stair(p, m, t, n, Q0, ub, 1, &c1);
            cells_static += c0; cells_narrow += c1;
            if (c1 < c0) narrowed_fewer++;
            if (s0 >= 0 && s0 != truth) { wrong++; if (wrong < 10) fprintf(stderr, "STATIC window wrong: case %ld m %d n %d Q0 %d ub %d: %d, distance %d\n", it, m, n, Q0, ub, s0, truth); }
            if (s1 >= 0 && s1 != truth) { wrong++; if (wrong < 10) fprintf(stderr, "NARROWING window wrong: case %ld kind %d m %d n %d Q0 %d ub %d: %d, distance %d\n", it, kind, m, n, Q0, ub, s1, truth); }
            if (s0 >= 0 && s1 < 0) { wrong++; if (wrong < 10) fprintf(stderr, "narrowing LOST an answer the static window gives: case %ld m %d n %d Q0 %d ub %d\n", it, m, n, Q0, ub); }
            if (s1 >= 0) accepted++; else refused++;
        }
    }
    printf("%ld pairs x 4 settings: %ld accepted, %ld refused, %ld wrong; narrowing computed fewer cells in %ld runs (%.3g of %.3g window cells)\n", cases, accepted, refused, wrong,
           narrowed_fewer, (double)cells_narrow, (double)cells_static);
    return wrong ? 1 : 0;
}
