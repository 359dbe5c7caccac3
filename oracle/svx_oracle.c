/*
 * svx_oracle.c - CPU restatement of SVIM's COLLECT+CLUSTER path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a plain-C, single-threaded restatement of the reference algorithm
 * (eldariont/svim v2.0.0, pure Python) with the same Structure-of-Arrays interface as the product's C ABI
 * (include/svx.h), so that tests can diff the HIP path against it array by array.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product (svim_amd/) never does.
 *
 * Pinning: the restatement is checked against golden vectors produced by RUNNING THE REFERENCE in the build
 * container (tests/golden/make_golden.py; tests/test_oracle_golden.py).  Third-party behaviour the reference
 * reaches through wheels is restated from the published algorithms and pinned the same way:
 *   scipy.cluster.hierarchy.linkage('average') / fcluster('distance')  (scipy 1.15.3)  -> svo_linkage_fcluster
 *   CPython random.seed / random.sample (3.10.12, MT19937)                             -> mt_* / svo_sample
 *   edlib.align(...)["editDistance"] (unpinned in setup.py:41; unit-cost global Levenshtein) -> svo_edit_distance
 *   statistics.mean / stdev -> FP64 restatement, equal to the reference within 1e-12 relative (the reference's
 *   exact-fraction arithmetic is not reproduced bit for bit; north_star tolerance for FP scores is 1e-6)
 *
 * Each function cites the reference lines it follows.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "../include/svx.h"

/* ---------------------------------------------------------------- growable signature table ---- */
typedef struct sigtab {
    int64_t n, cap;
    uint64_t* key; uint8_t* type; uint8_t* src; uint8_t* aux;
    int32_t *contig, *start, *end, *contig2, *pos2, *read_id;
    int32_t *rec, *qpos, *qlen;      /* INS: record holding SEQ, slice start, slice length */
} sigtab;

static void st_reserve(sigtab* t, int64_t need) {
    if (need <= t->cap) return;
    int64_t c = t->cap ? t->cap * 2 : 1024;
    while (c < need) c *= 2;
#define GROW(f) t->f = realloc(t->f, (size_t)c * sizeof(*t->f))
    GROW(key); GROW(type); GROW(src); GROW(aux); GROW(contig); GROW(start); GROW(end); GROW(contig2);
    GROW(pos2); GROW(read_id); GROW(rec); GROW(qpos); GROW(qlen);
#undef GROW
    t->cap = c;
}
static void st_free(sigtab* t) {
    free(t->key); free(t->type); free(t->src); free(t->aux); free(t->contig); free(t->start); free(t->end);
    free(t->contig2); free(t->pos2); free(t->read_id); free(t->rec); free(t->qpos); free(t->qlen);
    memset(t, 0, sizeof(*t));
}
static int64_t st_push(sigtab* t, uint64_t key, int type, int src, int aux, int32_t contig, int64_t start,
                       int64_t end, int32_t contig2, int64_t pos2, int32_t read_id) {
    st_reserve(t, t->n + 1);
    int64_t i = t->n++;
    t->key[i] = key; t->type[i] = (uint8_t)type; t->src[i] = (uint8_t)src; t->aux[i] = (uint8_t)aux;
    t->contig[i] = contig; t->start[i] = (int32_t)start; t->end[i] = (int32_t)end; t->contig2[i] = contig2;
    t->pos2[i] = (int32_t)pos2; t->read_id[i] = read_id; t->rec[i] = -1; t->qpos[i] = 0; t->qlen[i] = 0;
    return i;
}

typedef int (*svo_chain_fn)(void* user, int phase, int64_t* words /* [SVX_NTYPES] */);
typedef struct svo_ctx {
    sigtab sig, bnd;                 /* results of the last collect */
    int64_t* sig_seq_off; uint8_t* sig_seq;     /* INS sequences of `sig`, per signature */
    /* genome */
    int32_t g_n; int64_t* g_off; uint8_t* g_codes;
    /* cluster results */
    svx_cluster_view clu; int64_t clu_cap, mem_cap; int64_t* part_index;
    /* svo_cluster_set_chain: the checker's stand-in for the rank exchange of svx_cluster_set_ranks - fn(user, 0, words) OBTAINS the 6 stream start
     * positions of this rank (words earlier ranks' partitions consumed), fn(user, 1, words) REPORTS the 6 end positions */
    svo_chain_fn chain_fn; void* chain_user;
    svx_stats stats;
    int n_threads;                   /* svo_set_threads: worker threads for the pair distances of svo_cluster (default 1) */
} svo_ctx;

int svo_ctx_create(svo_ctx** out) { *out = calloc(1, sizeof(svo_ctx)); return 0; }
static void clu_free(svo_ctx* c) {
    svx_cluster_view* v = &c->clu;
    free(v->type); free(v->contig); free(v->start); free(v->end); free(v->contig2); free(v->start2); free(v->end2);
    free(v->aux); free(v->score); free(v->std_span); free(v->std_pos); free(v->size); free(v->member_off);
    free(v->members); free(c->part_index);
    memset(v, 0, sizeof(*v)); c->part_index = NULL; c->clu_cap = c->mem_cap = 0;
}
void svo_ctx_destroy(svo_ctx* c) {
    if (!c) return;
    st_free(&c->sig); st_free(&c->bnd); free(c->sig_seq_off); free(c->sig_seq);
    free(c->g_off); free(c->g_codes); clu_free(c); free(c);
}
int svo_get_stats(svo_ctx* c, svx_stats* out) { *out = c->stats; return 0; }

/* ================================================================= COLLECT ==================== */

#define KEY(slot, phase, ord) (((uint64_t)(slot) << 32) | ((uint64_t)(phase) << 30) | (uint64_t)(ord))

/* Python slice semantics seq[a:b] on a sequence of length len -> [lo, hi) */
static void py_slice(int64_t a, int64_t b, int64_t len, int64_t* lo, int64_t* hi) {
    if (a < 0) { a += len; if (a < 0) a = 0; } else if (a > len) a = len;
    if (b < 0) { b += len; if (b < 0) b = 0; } else if (b > len) b = len;
    if (b < a) b = a;
    *lo = a; *hi = b;
}

/* htslib-derived coordinates of one alignment from its packed CIGAR (SURVEY.md section 8 row a3;
 * pysam accessors used at src/svim/SVIM_inter.py:30-46). */
typedef struct geom { int64_t ref_len, qstart, qend, read_len, hard; int has_cigar; } geom;
static geom cigar_geom(const uint32_t* c, int64_t n, int64_t lseq) {
    geom g; memset(&g, 0, sizeof g);
    g.has_cigar = n > 0;
    int64_t lead_s = 0; int in_lead = 1;
    int64_t mix = 0;             /* M+I+=+X */
    for (int64_t k = 0; k < n; k++) {
        int op = c[k] & 15; int64_t l = c[k] >> 4;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) g.ref_len += l;
        if (op == 0 || op == 1 || op == 4 || op == 5 || op == 7 || op == 8) g.read_len += l;
        if (op == 0 || op == 1 || op == 7 || op == 8) mix += l;
        if (op == 5) g.hard += l;
        if (in_lead) { if (op == 5) {} else if (op == 4) lead_s += l; else in_lead = 0; }
    }
    g.qstart = lead_s;
    if (lseq == 0) {
        /* no stored sequence: M+I+=+X plus soft clips met while the running total is still 0 */
        int64_t e = 0;
        for (int64_t k = 0; k < n; k++) {
            int op = c[k] & 15; int64_t l = c[k] >> 4;
            if (op == 0 || op == 1 || op == 7 || op == 8 || (op == 4 && e == 0)) e += l;
        }
        g.qend = e;
    } else {
        int64_t e = lseq;
        for (int64_t k = n - 1; k >= 1; k--) {      /* element 0 is never inspected */
            int op = c[k] & 15; int64_t l = c[k] >> 4;
            if (op == 5) continue; else if (op == 4) e -= l; else break;
        }
        g.qend = e;
    }
    if (g.ref_len == 0) g.ref_len = 1;               /* bam_endpos: at least one base */
    (void)mix;
    return g;
}

/* analyze_cigar_indel + analyze_alignment_indel (src/svim/SVIM_intra.py:8-51) for one record */
static void collect_indels(svo_ctx* ctx, const svx_batch* b, const svx_params* p, int64_t r) {
    const uint32_t* c = b->cigar + b->cigar_off[r];
    int64_t n = (int64_t)(b->cigar_off[r + 1] - b->cigar_off[r]);
    int64_t pos_ref = 0, pos_read = 0;
    int64_t rs = b->pos[r];
    uint32_t slot = b->order[r];
    for (int64_t k = 0; k < n; k++) {
        int op = c[k] & 15; int64_t l = c[k] >> 4;
        if (op == 0 || op == 7 || op == 8) { pos_ref += l; pos_read += l; }
        else if (op == 1) {
            if (l >= p->min_sv_size) {
                int64_t i = st_push(&ctx->sig, KEY(slot, 0, k), SVX_INS, SVX_SRC_CIGAR, 0, b->tid[r], rs + pos_ref,
                                    rs + pos_ref + l, -1, 0, b->read_id[r]);
                int64_t lo, hi; py_slice(pos_read, pos_read + l, b->lseq[r], &lo, &hi);
                ctx->sig.rec[i] = (int32_t)r; ctx->sig.qpos[i] = (int32_t)lo; ctx->sig.qlen[i] = (int32_t)(hi - lo);
            }
            pos_read += l;
        } else if (op == 2) {
            if (l >= p->min_sv_size) {
                st_push(&ctx->sig, KEY(slot, 0, k), SVX_DEL, SVX_SRC_CIGAR, 0, b->tid[r], rs + pos_ref,
                        rs + pos_ref + l, -1, 0, b->read_id[r]);
                if (p->all_bnds)     /* SignatureTranslocation(chr, s, 'fwd', chr, e, 'fwd') : s < e so no swap */
                    st_push(&ctx->bnd, KEY(slot, 0, k), SVX_BND, SVX_SRC_CIGAR, 0, b->tid[r], rs + pos_ref,
                            rs + pos_ref + 1, b->tid[r], rs + pos_ref + l, b->read_id[r]);
            }
            pos_ref += l;
        } else if (op == 4) pos_read += l;
        /* N, H, P, B: ignored (N does not advance the reference cursor - faithful quirk) */
    }
}

/* SignatureTranslocation.__init__ canonical order (src/svim/SVSignature.py:194-211) */
static void push_bnd(sigtab* t, uint64_t key, const int32_t* rank, int32_t c1, int64_t p1, int rev1, int32_t c2,
                     int64_t p2, int rev2, int32_t read_id) {
    int keep = (rank[c1] < rank[c2]) || (c1 == c2 && p1 < p2);
    if (keep) st_push(t, key, SVX_BND, SVX_SRC_SUPPL, (rev1 ? 1 : 0) | (rev2 ? 2 : 0), c1, p1, p1 + 1, c2, p2, read_id);
    else      st_push(t, key, SVX_BND, SVX_SRC_SUPPL, (rev2 ? 0 : 1) | (rev1 ? 0 : 2), c2, p2, p2 + 1, c1, p1, read_id);
}

/* is_similar (src/svim/SVIM_inter.py:11-21) on doubles (means may be non-integral) */
static double py_floordiv2(double x) { return floor(x / 2.0); }
static int is_similar(int32_t chr1, double s1, double e1, int32_t chr2, double s2, double e2, double thr) {
    double span1 = e1 - s1, span2 = e2 - s2;
    double c1 = py_floordiv2(s1 + e1), c2 = py_floordiv2(s2 + e2);
    double pd = fabs(c1 - c2) / 900.0;
    double mx = span1 > span2 ? span1 : span2;
    double sd = fabs(span1 - span2) / mx;
    return chr1 == chr2 && pd + sd < thr;
}

typedef struct aseg { int64_t q_start, q_end, ref_start, ref_end; int32_t ref_id; int rev; } aseg;
typedef struct tdup { int32_t chr; int64_t s, e; int full, fwd; } tdup;
typedef struct trn { int d1, d2; int32_t c1; int64_t p1; int32_t c2; int64_t p2; } trn;   /* d: 0 fwd 1 rev */

/* analyze_read_segments (src/svim/SVIM_inter.py:24-302) for primary record r */
static void collect_segments(svo_ctx* ctx, const svx_batch* b, const svx_params* p, int64_t r) {
    int64_t s0 = b->seg_off[r], s1 = b->seg_off[r + 1];
    const uint32_t* pc = b->cigar + b->cigar_off[r];
    int64_t pn = (int64_t)(b->cigar_off[r + 1] - b->cigar_off[r]);
    geom pg = cigar_geom(pc, pn, b->lseq[r]);
    if ((b->flag[r] & SVX_FLAG_SA) && pg.hard > 0) s1 = s0;   /* retrieve_other_alignments returns [] (SVIM_COLLECT.py:47) */
    int64_t cap = (s1 - s0) + 1;
    aseg* al = malloc(sizeof(aseg) * (size_t)cap);
    int64_t na = 0;
    /* alignments = [primary] + supplementaries; reverse ones need infer_read_length (skip if None) */
    for (int64_t k = -1; k < s1 - s0; k++) {
        geom g; int rev; int32_t tid; int64_t pos; int usable;
        if (k < 0) { g = pg; rev = (b->flag[r] & 16) != 0; tid = b->tid[r]; pos = b->pos[r]; usable = 1; }
        else {
            int64_t s = s0 + k;
            if (b->seg_mapq[s] < p->min_mapq) continue;           /* good_suppl_alns filter, SVIM_COLLECT.py:154 */
            g = cigar_geom(b->seg_cigar + b->seg_cigar_off[s], (int64_t)(b->seg_cigar_off[s + 1] - b->seg_cigar_off[s]),
                           b->seg_lseq[s]);
            rev = b->seg_rev[s]; tid = b->seg_tid[s]; pos = b->seg_pos[s]; usable = 1;
        }
        aseg a;
        if (rev) {
            if (!g.has_cigar || g.read_len <= 0) continue;        /* infer_read_length() is None -> skipped */
            a.q_start = g.read_len - g.qend; a.q_end = g.read_len - g.qstart;
        } else { a.q_start = g.qstart; a.q_end = g.qend; }
        a.ref_id = tid; a.ref_start = pos; a.ref_end = pos + g.ref_len; a.rev = rev;
        (void)usable;
        al[na++] = a;
    }
    /* stable sort by (q_start, q_end) */
    for (int64_t i = 1; i < na; i++) {
        aseg x = al[i]; int64_t j = i - 1;
        while (j >= 0 && (al[j].q_start > x.q_start || (al[j].q_start == x.q_start && al[j].q_end > x.q_end))) { al[j + 1] = al[j]; j--; }
        al[j + 1] = x;
    }
    tdup* td = malloc(sizeof(tdup) * (size_t)(na + 1)); int64_t ntd = 0;
    trn* tr = malloc(sizeof(trn) * (size_t)(na + 1)); int64_t ntr = 0;
    const int32_t* rank = b->contig_rank;
    uint32_t slot = b->seg_order[r];
    int32_t rid = b->read_id[r];
    int64_t MIN = p->min_sv_size, MAX = p->max_sv_size, GAP = p->segment_gap_tolerance, OVL = p->segment_overlap_tolerance;
    int64_t prim_len = b->lseq[r];
    int64_t prim_infer = pg.read_len;         /* primary.infer_read_length() */
#define SIG(type, aux, c, s, e) st_push(&ctx->sig, KEY(slot, 1, idx), type, SVX_SRC_SUPPL, aux, c, s, e, -1, 0, rid)
#define BND_MAIN(c1, p1, r1, c2, p2, r2) do { push_bnd(&ctx->sig, KEY(slot, 1, idx), rank, c1, p1, r1, c2, p2, r2, rid); \
        trn t_ = { r1, r2, c1, p1, c2, p2 }; tr[ntr++] = t_; } while (0)
#define BND_SIDE(c1, p1, r1, c2, p2, r2) do { if (p->all_bnds) push_bnd(&ctx->bnd, KEY(slot, 1, idx), rank, c1, p1, r1, c2, p2, r2, rid); } while (0)
    for (int64_t idx = 0; idx + 1 < na; idx++) {
        aseg cu = al[idx], nx = al[idx + 1];
        int64_t dr = nx.q_start - cu.q_end;
        if (cu.ref_id == nx.ref_id) {
            int32_t chr = cu.ref_id;
            if (cu.rev == nx.rev) {
                int64_t dref = cu.rev ? cu.ref_start - nx.ref_end : nx.ref_start - cu.ref_end;
                if (dr >= -OVL) {
                    if (dref >= -OVL) {
                        int64_t dev = dr - dref;
                        if (dev >= MIN) {                                         /* INS candidate, :80-94 */
                            if (dref <= GAP) {
                                int64_t st = cu.rev ? cu.ref_start : cu.ref_end;
                                int64_t a = cu.rev ? prim_infer - nx.q_start : cu.q_end;
                                int64_t i = SIG(SVX_INS, 0, chr, st, st + dev);
                                int64_t lo, hi;
                                if (prim_len > 0) py_slice(a, a + dev, prim_len, &lo, &hi); else { lo = hi = 0; }
                                ctx->sig.rec[i] = (int32_t)r; ctx->sig.qpos[i] = (int32_t)lo; ctx->sig.qlen[i] = (int32_t)(hi - lo);
                            }
                        } else if (-MAX <= dev && dev <= -MIN) {                  /* DEL candidate, :96-106 */
                            if (dr <= GAP) {
                                int64_t st = cu.rev ? nx.ref_end : cu.ref_end;
                                SIG(SVX_DEL, 0, chr, st, st - dev);
                                BND_SIDE(chr, st - 1, 0, chr, st - dev, 0);
                            }
                        } else if (dev < -MAX) {                                  /* very large DEL or TRANS, :108-116 */
                            if (dr <= GAP) {
                                if (!cu.rev) BND_MAIN(chr, cu.ref_end - 1, 0, chr, nx.ref_start, 0);
                                else         BND_MAIN(chr, cu.ref_start, 1, chr, nx.ref_end - 1, 1);
                            }
                        }
                    } else if (dref <= -MIN) {                                    /* overlap on reference, :118-150 */
                        if (!cu.rev) {
                            if (nx.ref_end > cu.ref_start) {
                                tdup t = { chr, nx.ref_start, cu.ref_end, 1, 1 }; td[ntd++] = t;
                                BND_SIDE(chr, cu.ref_end - 1, 0, chr, nx.ref_start, 0);
                            } else if (dref >= -MAX) {
                                tdup t = { chr, nx.ref_start, cu.ref_end, 0, 1 }; td[ntd++] = t;
                                BND_SIDE(chr, cu.ref_end - 1, 0, chr, nx.ref_start, 0);
                            } else BND_MAIN(chr, cu.ref_end - 1, 0, chr, nx.ref_start, 0);
                        } else {
                            if (nx.ref_start < cu.ref_end) {
                                tdup t = { chr, cu.ref_start, nx.ref_end, 1, 0 }; td[ntd++] = t;
                                BND_SIDE(chr, cu.ref_start, 1, chr, nx.ref_end - 1, 1);
                            } else if (dref >= -MAX) {
                                tdup t = { chr, cu.ref_start, nx.ref_end, 0, 0 }; td[ntd++] = t;
                                BND_SIDE(chr, cu.ref_start, 1, chr, nx.ref_end - 1, 1);
                            } else BND_MAIN(chr, cu.ref_start, 1, chr, nx.ref_end - 1, 1);
                        }
                    }
                }
            } else if (!cu.rev && nx.rev) {                                       /* normal -> reverse, :154-178 */
                if (-OVL <= dr && dr <= GAP) {
                    if (nx.ref_start - cu.ref_end >= -OVL) {                      /* case 1 */
                        int64_t sz = nx.ref_end - cu.ref_end;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_LEFT_FWD, chr, cu.ref_end, nx.ref_end);
                            BND_SIDE(chr, cu.ref_end - 1, 0, chr, nx.ref_end - 1, 1); }
                        else if (sz > MAX) BND_MAIN(chr, cu.ref_end - 1, 0, chr, nx.ref_end - 1, 1);
                    } else if (cu.ref_start - nx.ref_end >= -OVL) {               /* case 3 */
                        int64_t sz = cu.ref_end - nx.ref_end;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_LEFT_REV, chr, nx.ref_end, cu.ref_end);
                            BND_SIDE(chr, cu.ref_end - 1, 0, chr, nx.ref_end - 1, 1); }
                        else if (sz > MAX) BND_MAIN(chr, cu.ref_end - 1, 0, chr, nx.ref_end - 1, 1);
                    }
                }
            } else {                                                              /* reverse -> normal, :180-204 */
                if (-OVL <= dr && dr <= GAP) {
                    if (nx.ref_start - cu.ref_end >= -OVL) {                      /* case 2 */
                        int64_t sz = nx.ref_start - cu.ref_start;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_RIGHT_FWD, chr, cu.ref_start, nx.ref_start);
                            BND_SIDE(chr, cu.ref_start, 1, chr, nx.ref_start, 0); }
                        else if (sz > MAX) BND_MAIN(chr, cu.ref_start, 1, chr, nx.ref_start, 0);
                    } else if (cu.ref_start - nx.ref_end >= -OVL) {               /* case 4 */
                        int64_t sz = cu.ref_start - nx.ref_start;
                        if (MIN <= sz && sz <= MAX) { SIG(SVX_INV, SVX_RIGHT_REV, chr, nx.ref_start, cu.ref_start);
                            BND_SIDE(chr, cu.ref_start, 1, chr, nx.ref_start, 0); }
                        else if (sz > MAX) BND_MAIN(chr, cu.ref_start, 1, chr, nx.ref_start, 0);
                    }
                }
            }
        } else {                                                                  /* different chromosomes, :206-240 */
            if (dr >= -OVL && dr <= GAP) {
                if (cu.rev == nx.rev) {
                    if (!cu.rev) BND_MAIN(cu.ref_id, cu.ref_end - 1, 0, nx.ref_id, nx.ref_start, 0);
                    else         BND_MAIN(cu.ref_id, cu.ref_start, 1, nx.ref_id, nx.ref_end - 1, 1);
                } else {
                    if (!cu.rev) BND_MAIN(cu.ref_id, cu.ref_end - 1, 0, nx.ref_id, nx.ref_end - 1, 1);
                    else         BND_MAIN(cu.ref_id, cu.ref_start, 1, nx.ref_id, nx.ref_start, 0);
                }
            }
        }
    }
#undef SIG
#undef BND_MAIN
#undef BND_SIDE
    /* tandem duplication runs (:242-272); note the stale current_direction after the first run */
    if (ntd > 0) {
        int32_t cur_chr = td[0].chr; int64_t sum_s = td[0].s, sum_e = td[0].e, cnt = 1; int any_full = td[0].full;
        int cur_dir = td[0].fwd; uint32_t ord = 0;
        for (int64_t k = 1; k <= ntd; k++) {
            int merge = 0;
            if (k < ntd) {
                double ms = (double)sum_s / (double)cnt, me = (double)sum_e / (double)cnt;
                merge = is_similar(cur_chr, ms, me, td[k].chr, (double)td[k].s, (double)td[k].e, 0.3) && cur_dir == td[k].fwd;
            }
            if (merge) { sum_s += td[k].s; sum_e += td[k].e; cnt++; any_full |= td[k].full; }
            else {
                /* int(mean(..)): exact rational truncated toward zero */
                int64_t ms = sum_s / cnt, me = sum_e / cnt;
                st_push(&ctx->sig, KEY(slot, 2, ord++), SVX_DUP_TAN, SVX_SRC_SUPPL, any_full ? 1 : 0, cur_chr, ms, me, -1, cnt, rid);
                if (k < ntd) { cur_chr = td[k].chr; sum_s = td[k].s; sum_e = td[k].e; cnt = 1; any_full = td[k].full; }
            }
        }
    }
    /* insertions_from (:274-300) */
    for (int64_t ti = 0; ti < ntr; ti++) {
        trn t = tr[ti];
        for (int64_t bi = 0; bi < ti; bi++) {
            trn q = tr[bi];
            if (q.d1 == t.d2 && q.d2 == t.d1 &&
                is_similar(q.c1, (double)q.p1, (double)(q.p1 + 1), t.c2, (double)t.p2, (double)(t.p2 + 1), 0.1) &&
                q.c2 == t.c1 && q.d2 == q.d1) {
                uint64_t key = KEY(slot, 3, (uint64_t)ti * (uint64_t)(ntr) + (uint64_t)bi);
                if (q.d1 == 0) {
                    int64_t sz = t.p1 - q.p2 + 1;
                    if (MIN <= sz && sz <= MAX)
                        st_push(&ctx->sig, key, SVX_DUP_INT, SVX_SRC_SUPPL, 0, q.c2, q.p2, t.p1 + 1, q.c1, (q.p1 + 1 + t.p2) / 2, rid);
                } else {
                    int64_t sz = q.p2 - t.p1;
                    if (MIN <= sz && sz <= MAX)
                        st_push(&ctx->sig, key, SVX_DUP_INT, SVX_SRC_SUPPL, 0, q.c2, t.p1, q.p2 + 1, q.c1, (q.p1 + t.p2 + 1) / 2, rid);
                }
            }
        }
    }
    free(al); free(td); free(tr);
}

/* stable sort of a sigtab by key (keys are unique) */
static int cmp_key_idx(const void* a, const void* b, void* keys) {
    const uint64_t* k = keys; int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return k[x] < k[y] ? -1 : (k[x] > k[y] ? 1 : (x < y ? -1 : (x > y)));
}
static void st_sort(sigtab* t) {
    int64_t n = t->n; if (n < 2) return;
    int64_t* idx = malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; i++) idx[i] = i;
    qsort_r(idx, (size_t)n, sizeof(int64_t), cmp_key_idx, t->key);
#define PERM(f) do { __typeof__(t->f) tmp = malloc((size_t)n * sizeof(*t->f)); for (int64_t i = 0; i < n; i++) tmp[i] = t->f[idx[i]]; \
        memcpy(t->f, tmp, (size_t)n * sizeof(*t->f)); free(tmp); } while (0)
    PERM(key); PERM(type); PERM(src); PERM(aux); PERM(contig); PERM(start); PERM(end); PERM(contig2); PERM(pos2);
    PERM(read_id); PERM(rec); PERM(qpos); PERM(qlen);
#undef PERM
    free(idx);
}

/* analyze_alignment_file_coordsorted / _querysorted (src/svim/SVIM_COLLECT.py:96-167).  Record filtering
 * as at :143 (and :108 via SVX_FLAG_SKIP); supplementary records contribute indels only (:145-148). */
int svo_collect(svo_ctx* ctx, const svx_batch* b, const svx_params* p) {
    ctx->sig.n = 0; ctx->bnd.n = 0;
    memset(&ctx->stats, 0, sizeof ctx->stats);
    for (int64_t r = 0; r < b->n_rec; r++) {
        uint16_t f = b->flag[r];
        if ((f & SVX_FLAG_SKIP) || (f & 4) || (f & 256) || b->mapq[r] < p->min_mapq) continue;
        ctx->stats.n_rec_used++;
        ctx->stats.n_ops += (int64_t)(b->cigar_off[r + 1] - b->cigar_off[r]);
        collect_indels(ctx, b, p, r);
        if (!(f & 2048)) collect_segments(ctx, b, p, r);
    }
    st_sort(&ctx->sig); st_sort(&ctx->bnd);
    /* materialise inserted sequences: query_sequence[pos_read:pos_read+len] (SVIM_intra.py:47, SVIM_inter.py:85,91) */
    sigtab* t = &ctx->sig;
    free(ctx->sig_seq_off); free(ctx->sig_seq);
    ctx->sig_seq_off = malloc(sizeof(int64_t) * (size_t)(t->n + 1));
    int64_t tot = 0;
    for (int64_t i = 0; i < t->n; i++) { ctx->sig_seq_off[i] = tot; tot += t->qlen[i]; }
    ctx->sig_seq_off[t->n] = tot;
    ctx->sig_seq = malloc((size_t)(tot ? tot : 1));
    for (int64_t i = 0; i < t->n; i++) {
        if (!t->qlen[i]) continue;
        const uint8_t* s = b->seq + b->seq_off[t->rec[i]];
        int64_t q0 = t->qpos[i];
        if (b->seq_rng_off) {                              /* sparse SEQ (svx_batch.seq_rng_*): the range that holds the insertion */
            s = NULL;
            for (uint32_t r = b->seq_rng_off[t->rec[i]]; r < b->seq_rng_off[t->rec[i] + 1]; r++)
                if (b->seq_rng_q0[r] <= q0 && q0 + t->qlen[i] <= (int64_t)b->seq_rng_q0[r] + b->seq_rng_len[r]) { s = b->seq + b->seq_rng_byte[r]; q0 -= b->seq_rng_q0[r]; break; }
            if (!s) return -3;
        }
        uint8_t* o = ctx->sig_seq + ctx->sig_seq_off[i];
        for (int64_t k = 0; k < t->qlen[i]; k++) { int64_t q = q0 + k; uint8_t by = s[q >> 1]; o[k] = (q & 1) ? (by & 15) : (by >> 4); }
    }
    ctx->stats.n_sig = t->n; ctx->stats.n_bnd_side = ctx->bnd.n; ctx->stats.n_ins_bases = tot;
    return 0;
}
int svo_collect_count(svo_ctx* ctx, int64_t* n_sig, int64_t* n_seq, int64_t* n_bnd) {
    *n_sig = ctx->sig.n; *n_seq = ctx->sig_seq_off ? ctx->sig_seq_off[ctx->sig.n] : 0; *n_bnd = ctx->bnd.n; return 0;
}
int svo_collect_fetch(svo_ctx* ctx, int which, svx_sig_view* o) {
    sigtab* t = which ? &ctx->bnd : &ctx->sig;
    size_t n = (size_t)t->n;
    if (n) {                                               /* (an empty table has no arrays: nothing to copy from) */
        memcpy(o->key, t->key, n * 8); memcpy(o->type, t->type, n); memcpy(o->src, t->src, n); memcpy(o->aux, t->aux, n);
        memcpy(o->contig, t->contig, n * 4); memcpy(o->start, t->start, n * 4); memcpy(o->end, t->end, n * 4);
        memcpy(o->contig2, t->contig2, n * 4); memcpy(o->pos2, t->pos2, n * 4); memcpy(o->read_id, t->read_id, n * 4);
    }
    if (which == 0 && ctx->sig_seq_off) {
        memcpy(o->seq_off, ctx->sig_seq_off, (n + 1) * 8);
        if (ctx->sig_seq && ctx->sig_seq_off[n]) memcpy(o->seq, ctx->sig_seq, (size_t)ctx->sig_seq_off[n]);      /* (no inserted bases: no array, nothing to copy from) */
    } else { for (size_t i = 0; i <= n; i++) o->seq_off[i] = 0; }
    o->n = t->n;
    return 0;
}

/* analyze_cigar_indel alone (src/svim/SVIM_intra.py:8-30) */
int svo_cigar_indel(const uint32_t* c, int64_t n, int32_t min_length, int64_t* o_ref, int64_t* o_read, int32_t* o_len,
                    uint8_t* o_del, int64_t* o_n) {
    int64_t pr = 0, pq = 0, m = 0;
    for (int64_t k = 0; k < n; k++) {
        int op = c[k] & 15; int64_t l = c[k] >> 4;
        if (op == 0 || op == 7 || op == 8) { pr += l; pq += l; }
        else if (op == 1) { if (l >= min_length) { o_ref[m] = pr; o_read[m] = pq; o_len[m] = (int32_t)l; o_del[m] = 0; m++; } pq += l; }
        else if (op == 2) { if (l >= min_length) { o_ref[m] = pr; o_read[m] = pq; o_len[m] = (int32_t)l; o_del[m] = 1; m++; } pr += l; }
        else if (op == 4) pq += l;
    }
    *o_n = m; return 0;
}

/* ================================================================= CLUSTER ==================== */

/* ---- MT19937 as CPython uses it (random.seed(int) -> init_by_array; Modules/_randommodule.c) ---- */
typedef struct mt { uint32_t s[624]; int idx; int64_t drawn; } mt;
static void mt_init_genrand(mt* m, uint32_t s) {
    m->s[0] = s;
    for (int i = 1; i < 624; i++) m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624; m->drawn = 0;
}
static void mt_seed_int(mt* m, uint32_t key0) {            /* init_by_array(key=[key0], 1) */
    mt_init_genrand(m, 19650218u);
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
        m->s[i] = (m->s[i] ^ ((m->s[i - 1] ^ (m->s[i - 1] >> 30)) * 1664525u)) + key0 + (uint32_t)j;
        i++; j++; if (i >= 624) { m->s[0] = m->s[623]; i = 1; } if (j >= 1) j = 0;
    }
    for (int k = 623; k; k--) {
        m->s[i] = (m->s[i] ^ ((m->s[i - 1] ^ (m->s[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++; if (i >= 624) { m->s[0] = m->s[623]; i = 1; }
    }
    m->s[0] = 0x80000000u; m->idx = 624;
}
static uint32_t mt_u32(mt* m) {
    if (m->idx >= 624) {
        uint32_t* s = m->s; int kk;
        for (kk = 0; kk < 624 - 397; kk++) { uint32_t y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7fffffffu); s[kk] = s[kk + 397] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
        for (; kk < 623; kk++) { uint32_t y = (s[kk] & 0x80000000u) | (s[kk + 1] & 0x7fffffffu); s[kk] = s[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
        uint32_t y = (s[623] & 0x80000000u) | (s[0] & 0x7fffffffu); s[623] = s[396] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
        m->idx = 0;
    }
    uint32_t y = m->s[m->idx++];
    m->drawn++;
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
static int bit_length(uint32_t n) { int k = 0; while (n) { k++; n >>= 1; } return k; }
static uint32_t mt_randbelow(mt* m, uint32_t n) {           /* Random._randbelow_with_getrandbits */
    int k = bit_length(n);
    uint32_t r = mt_u32(m) >> (32 - k);
    while (r >= n) r = mt_u32(m) >> (32 - k);
    return r;
}
/* random.sample(range(n), 100) index list (Lib/random.py sample(); k = 100 -> setsize = 1045) */
static void mt_sample100(mt* m, int64_t n, int32_t* out) {
    const int k = 100;
    if (n <= 1045) {
        int32_t pool[1045];
        for (int64_t i = 0; i < n; i++) pool[i] = (int32_t)i;
        for (int i = 0; i < k; i++) { uint32_t j = mt_randbelow(m, (uint32_t)(n - i)); out[i] = pool[j]; pool[j] = pool[n - i - 1]; }
    } else {
        for (int i = 0; i < k; i++) {
            uint32_t j;
            for (;;) { j = mt_randbelow(m, (uint32_t)n); int seen = 0; for (int q = 0; q < i; q++) if ((uint32_t)out[q] == j) { seen = 1; break; } if (!seen) break; }
            out[i] = (int32_t)j;
        }
    }
}
/* test hooks */
int svo_getrandbits(uint32_t seed, int k, int64_t count, uint32_t* out) {
    mt m; mt_seed_int(&m, seed);
    for (int64_t i = 0; i < count; i++) out[i] = mt_u32(&m) >> (32 - k);
    return 0;
}
int svo_sample_sequence(uint32_t seed, int64_t n_calls, const int64_t* ns, int32_t* out /* n_calls*100 */) {
    mt m; mt_seed_int(&m, seed);
    for (int64_t c = 0; c < n_calls; c++) mt_sample100(&m, ns[c], out + c * 100);
    return 0;
}

/* ---- unit-cost global edit distance (what edlib.align(a,b)["editDistance"] returns, SVIM_clustering.py:45):
 * Myers/Hyyro bit-vector, 64-bit blocks, full matrix ---- */
int32_t svo_edit_distance(const uint8_t* a, int64_t la, const uint8_t* b, int64_t lb) {
    if (la < lb) { const uint8_t* t = a; a = b; b = t; int64_t tl = la; la = lb; lb = tl; }
    if (lb == 0) return (int32_t)la;
    int64_t nb = (lb + 63) / 64;
    uint64_t* peq = calloc((size_t)(16 * nb), 8);
    uint64_t* pv = malloc((size_t)nb * 8); uint64_t* mv = calloc((size_t)nb, 8);
    for (int64_t i = 0; i < lb; i++) peq[(int64_t)(b[i] & 15) * nb + (i >> 6)] |= 1ull << (i & 63);
    for (int64_t k = 0; k < nb; k++) pv[k] = ~0ull;
    int64_t score = lb;
    int last_bit = (int)((lb - 1) & 63);
    for (int64_t j = 0; j < la; j++) {
        const uint64_t* eqv = peq + (int64_t)(a[j] & 15) * nb;
        int hin = 1;                                   /* global alignment: top row delta is +1 */
        for (int64_t k = 0; k < nb; k++) {
            uint64_t eq = eqv[k], PV = pv[k], MV = mv[k];
            uint64_t hin_neg = hin < 0 ? 1ull : 0ull;
            uint64_t xv = eq | MV;
            eq |= hin_neg;
            uint64_t xh = (((eq & PV) + PV) ^ PV) | eq;
            uint64_t ph = MV | ~(xh | PV);
            uint64_t mh = PV & xh;
            int hout = 0;
            int top = (k == nb - 1) ? last_bit : 63;
            if (ph & (1ull << top)) hout = 1; else if (mh & (1ull << top)) hout = -1;
            ph <<= 1; mh <<= 1;
            if (hin < 0) mh |= 1ull; else if (hin > 0) ph |= 1ull;
            pv[k] = mh | ~(xv | ph);
            mv[k] = ph & xv;
            hin = hout;
        }
        score += hin;
    }
    free(peq); free(pv); free(mv);
    return (int32_t)score;
}

/* ---- scipy linkage(method='average') + fcluster(criterion='distance') (SVIM_clustering.py:170-171;
 * scipy/cluster/_hierarchy.pyx nn_chain, label, cluster_dist) ---- */
static inline int64_t cidx(int64_t n, int64_t i, int64_t j) { if (i > j) { int64_t t = i; i = j; j = t; } return n * i - i * (i + 1) / 2 + (j - i - 1); }

int svo_linkage_fcluster(int32_t n, const double* d_in, double cutoff, int32_t* labels, double* Zout /* (n-1)*4 or NULL */) {
    if (n == 1) { labels[0] = 1; return 0; }
    int64_t np_ = (int64_t)n * (n - 1) / 2;
    double* D = malloc((size_t)np_ * 8); memcpy(D, d_in, (size_t)np_ * 8);
    int* size = malloc(sizeof(int) * (size_t)n); int* chain = malloc(sizeof(int) * (size_t)n);
    double* Z = malloc(sizeof(double) * 4 * (size_t)(n - 1));
    for (int i = 0; i < n; i++) size[i] = 1;
    int chain_len = 0, x = 0, y = 0;
    for (int k = 0; k < n - 1; k++) {
        double cur = 0;
        if (chain_len == 0) { chain_len = 1; for (int i = 0; i < n; i++) if (size[i] > 0) { chain[0] = i; break; } }
        for (;;) {
            x = chain[chain_len - 1];
            if (chain_len > 1) { y = chain[chain_len - 2]; cur = D[cidx(n, x, y)]; } else cur = INFINITY;
            for (int i = 0; i < n; i++) {
                if (size[i] == 0 || x == i) continue;
                double dist = D[cidx(n, x, i)];
                if (dist < cur) { cur = dist; y = i; }
            }
            if (chain_len > 1 && y == chain[chain_len - 2]) break;
            chain[chain_len++] = y;
        }
        chain_len -= 2;
        if (x > y) { int t = x; x = y; y = t; }
        int nx = size[x], ny = size[y];
        Z[4 * k] = x; Z[4 * k + 1] = y; Z[4 * k + 2] = cur; Z[4 * k + 3] = nx + ny;
        size[x] = 0; size[y] = nx + ny;
        for (int i = 0; i < n; i++) {
            if (size[i] == 0 || i == y) continue;
            D[cidx(n, i, y)] = ((double)nx * D[cidx(n, i, x)] + (double)ny * D[cidx(n, i, y)]) / (double)(nx + ny);
        }
    }
    /* stable sort of merges by height (np.argsort(kind='mergesort')) */
    int* ord = malloc(sizeof(int) * (size_t)(n - 1));
    for (int i = 0; i < n - 1; i++) ord[i] = i;
    for (int i = 1; i < n - 1; i++) { int v = ord[i]; int j = i - 1; while (j >= 0 && Z[4 * ord[j] + 2] > Z[4 * v + 2]) { ord[j + 1] = ord[j]; j--; } ord[j + 1] = v; }
    double* Zs = malloc(sizeof(double) * 4 * (size_t)(n - 1));
    for (int i = 0; i < n - 1; i++) memcpy(Zs + 4 * i, Z + 4 * ord[i], 32);
    /* label(): union-find relabel */
    int tot = 2 * n - 1;
    int* parent = malloc(sizeof(int) * (size_t)tot); int* usz = malloc(sizeof(int) * (size_t)tot);
    for (int i = 0; i < tot; i++) { parent[i] = i; usz[i] = i < n ? 1 : 0; }
    int next = n;
    for (int i = 0; i < n - 1; i++) {
        int a = (int)Zs[4 * i], b = (int)Zs[4 * i + 1];
        int ra = a; while (parent[ra] != ra) ra = parent[ra];
        int rb = b; while (parent[rb] != rb) rb = parent[rb];
        if (ra < rb) { Zs[4 * i] = ra; Zs[4 * i + 1] = rb; } else { Zs[4 * i] = rb; Zs[4 * i + 1] = ra; }
        parent[ra] = next; parent[rb] = next; usz[next] = usz[ra] + usz[rb]; Zs[4 * i + 3] = usz[next]; next++;
    }
    if (Zout) memcpy(Zout, Zs, sizeof(double) * 4 * (size_t)(n - 1));
    /* get_max_dist_for_each_cluster: children always have smaller row index than the parent row */
    double* MD = malloc(sizeof(double) * (size_t)(n - 1));
    for (int i = 0; i < n - 1; i++) {
        double m = Zs[4 * i + 2];
        int l = (int)Zs[4 * i], r = (int)Zs[4 * i + 1];
        if (l >= n && MD[l - n] > m) m = MD[l - n];
        if (r >= n && MD[r - n] > m) m = MD[r - n];
        MD[i] = m;
    }
    /* cluster_monocrit */
    int* stack = malloc(sizeof(int) * (size_t)n); unsigned char* vis = calloc((size_t)tot, 1);
    int k = 0, ncl = 0, leader = -1; stack[0] = 2 * n - 2;
    while (k >= 0) {
        int root = stack[k] - n;
        int lc = (int)Zs[4 * root], rc = (int)Zs[4 * root + 1];
        if (leader == -1 && MD[root] <= cutoff) { leader = root; ncl++; }
        if (lc >= n && !vis[lc]) { vis[lc] = 1; stack[++k] = lc; continue; }
        if (rc >= n && !vis[rc]) { vis[rc] = 1; stack[++k] = rc; continue; }
        if (lc < n) { if (leader == -1) ncl++; labels[lc] = ncl; }
        if (rc < n) { if (leader == -1) ncl++; labels[rc] = ncl; }
        if (leader == root) leader = -1;
        k--;
    }
    free(D); free(size); free(chain); free(Z); free(ord); free(Zs); free(parent); free(usz); free(MD); free(stack); free(vis);
    return 0;
}

/* ---- signature accessors on the SoA table (get_source / get_destination, SVSignature.py) ---- */
typedef struct csig {          /* one signature, decoded */
    int type, aux; int32_t contig, contig2, read_id; int64_t start, end, pos2;
    const uint8_t* seq; int64_t seq_len;
} csig;
static csig get_sig(const svx_sig_view* v, int64_t i) {
    csig s; s.type = v->type[i]; s.aux = v->aux[i]; s.contig = v->contig[i]; s.contig2 = v->contig2[i];
    s.read_id = v->read_id[i]; s.start = v->start[i]; s.end = v->end[i]; s.pos2 = v->pos2[i];
    if (v->seq_off) { s.seq = v->seq + v->seq_off[i]; s.seq_len = v->seq_off[i + 1] - v->seq_off[i]; } else { s.seq = NULL; s.seq_len = 0; }
    return s;
}
/* destination start: DUP_INT pos, BND pos2 (get_destination()[1]) */

/* reference.fetch(contig, max(0,a), max(0,b)) with end clipping */
static int64_t fetch(const svo_ctx* c, int32_t contig, int64_t a, int64_t b, uint8_t* out) {
    if (a < 0) a = 0;
    if (b < 0) b = 0;
    int64_t len = c->g_off[contig + 1] - c->g_off[contig];
    if (b > len) b = len;
    if (a >= b) return 0;
    memcpy(out, c->g_codes + c->g_off[contig] + a, (size_t)(b - a));
    return b - a;
}

/* compute_haplotype_edit_distance (src/svim/SVIM_clustering.py:32-45) */
static int64_t haplotype_edit_distance(svo_ctx* c, const csig* s1, const csig* s2) {
    int64_t ws = (s1->start < s2->start ? s1->start : s2->start) - 100;
    int64_t we = (s1->start > s2->start ? s1->start : s2->start) + 100;
    int64_t cap1 = (we - ws) + s1->seq_len + 8, cap2 = (we - ws) + s2->seq_len + 8;
    uint8_t* h1 = malloc((size_t)cap1); uint8_t* h2 = malloc((size_t)cap2);
    int64_t l1 = fetch(c, s1->contig, ws, s1->start, h1);
    memcpy(h1 + l1, s1->seq, (size_t)s1->seq_len); l1 += s1->seq_len;
    l1 += fetch(c, s1->contig, s1->start, we, h1 + l1);
    int64_t l2 = fetch(c, s2->contig, ws, s2->start, h2);
    memcpy(h2 + l2, s2->seq, (size_t)s2->seq_len); l2 += s2->seq_len;
    l2 += fetch(c, s2->contig, s2->start, we, h2 + l2);
    int64_t d = svo_edit_distance(h1, l1, h2, l2);
    __atomic_fetch_add(&c->stats.n_edit_pairs, 1, __ATOMIC_RELAXED); __atomic_fetch_add(&c->stats.n_edit_cells, l1 * l2, __ATOMIC_RELAXED);      /* (svo_set_threads) */
    free(h1); free(h2);
    return d;
}

static inline int64_t floordiv2(int64_t x) { return (x >= 0) ? x / 2 : -((-x + 1) / 2); }

/* span_position_distance (src/svim/SVIM_clustering.py:47-96).  Every '/' is one IEEE-754 FP64 division. */
static double span_position_distance(svo_ctx* c, const csig* a, const csig* b, const svx_params* p) {
    int t = a->type;
    if (t == SVX_BND) {
        int64_t d1 = llabs(a->start - b->start), d2 = llabs(a->pos2 - b->pos2);
        if (a->aux == b->aux) return (double)(d1 + d2) / 3000.0;
        return 99999.0;
    }
    int64_t span1 = a->end - a->start, span2 = b->end - b->start;
    int64_t mx = span1 > span2 ? span1 : span2;
    if (t == SVX_DEL || t == SVX_DUP_TAN || t == SVX_INV) {
        int64_t c1 = floordiv2(a->start + a->end), c2 = floordiv2(b->start + b->end);
        double pd = (double)llabs(c1 - c2) / p->position_distance_normalizer;
        double sd = (double)llabs(span1 - span2) / (double)mx;
        return pd + sd;
    }
    if (t == SVX_INS) {
        double pd = (double)llabs(a->start - b->start) / p->position_distance_normalizer;
        if (pd > 2 * p->cluster_max_distance) {
            double sd = (double)llabs(span1 - span2) / (double)mx;
            return pd + sd;
        }
        int64_t ed = haplotype_edit_distance(c, a, b);
        double sq = (double)ed / (double)mx / p->edit_distance_normalizer;
        return pd + sq;
    }
    /* DUP_INT */
    int64_t c1 = floordiv2(a->start + a->end), c2 = floordiv2(b->start + b->end);
    double pds = (double)llabs(c1 - c2) / p->position_distance_normalizer;
    double pdd = (double)llabs(a->pos2 - b->pos2) / p->position_distance_normalizer;
    double sd = (double)llabs(span1 - span2) / (double)mx;
    return pds + pdd + sd;
}

int svo_span_position_distance(svo_ctx* c, const svx_sig_view* v, int64_t i, int64_t j, const svx_params* p, double* out) {
    csig a = get_sig(v, i), b = get_sig(v, j);
    *out = span_position_distance(c, &a, &b, p);
    return 0;
}

int svo_set_genome(svo_ctx* c, const svx_genome* g) {
    free(c->g_off); free(c->g_codes);
    c->g_n = g->n_contig;
    c->g_off = malloc(sizeof(int64_t) * (size_t)(g->n_contig + 1));
    memcpy(c->g_off, g->off, sizeof(int64_t) * (size_t)(g->n_contig + 1));
    int64_t tot = g->off[g->n_contig];
    c->g_codes = malloc((size_t)(tot ? tot : 1));
    memcpy(c->g_codes, g->codes, (size_t)tot);
    return 0;
}

/* sort key of a signature (get_key, SVSignature.py:21-23,70-72,132-135,232-233) */
typedef struct skey { int32_t type, r1, r2; int64_t coord; int64_t idx; } skey;
static int cmp_skey(const void* a, const void* b) {
    const skey* x = a; const skey* y = b;
    if (x->type != y->type) return x->type < y->type ? -1 : 1;
    if (x->r1 != y->r1) return x->r1 < y->r1 ? -1 : 1;
    if (x->r2 != y->r2) return x->r2 < y->r2 ? -1 : 1;
    if (x->coord != y->coord) return x->coord < y->coord ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

/* mean / stdev in FP64 (see header note) */
static double fmean_i(const double* x, int n) { double s = 0; for (int i = 0; i < n; i++) s += x[i]; return s / (double)n; }
static double stdev_d(const double* x, int n) {
    double c = fmean_i(x, n);
    double ss = 0, sd = 0;
    for (int i = 0; i < n; i++) { double d = x[i] - c; ss += d * d; sd += d; }
    ss -= sd * sd / (double)n;
    if (ss < 0) ss = 0;
    return sqrt(ss / (double)(n - 1));
}
static double py_round_half_even(double x) { return nearbyint(x); }   /* default rounding mode = ties-to-even */

static void clu_reserve(svo_ctx* c, int64_t ncl, int64_t nmem) {
    svx_cluster_view* v = &c->clu;
    if (ncl > c->clu_cap) {
        int64_t k = c->clu_cap ? c->clu_cap * 2 : 256; while (k < ncl) k *= 2;
#define G(f) v->f = realloc(v->f, (size_t)k * sizeof(*v->f))
        G(type); G(contig); G(start); G(end); G(contig2); G(start2); G(end2); G(aux); G(score); G(std_span); G(std_pos); G(size);
#undef G
        v->member_off = realloc(v->member_off, (size_t)(k + 1) * 8);
        c->part_index = realloc(c->part_index, (size_t)k * 8);
        c->clu_cap = k;
    }
    if (nmem > c->mem_cap) { int64_t k = c->mem_cap ? c->mem_cap * 2 : 1024; while (k < nmem) k *= 2; v->members = realloc(v->members, (size_t)k * 4); c->mem_cap = k; }
}

/* calculate_score (src/svim/SVIM_clustering.py:183-211) */
static double calc_score(const csig* m, int n, int has_std, double std_span, double std_pos, double span, int type) {
    double sds = 0, pds = 0;
    if (has_std) {
        double a = std_span / span; sds = 1 - (a < 1 ? a : 1);      /* min(1, x) */
        double b = std_pos / span;  pds = 1 - (b < 1 ? b : 1);
    }
    int num;
    if (type == SVX_INV) {
        int cnt[5] = { 0, 0, 0, 0, 0 };
        for (int i = 0; i < n; i++) if (m[i].aux >= 0 && m[i].aux < 5) cnt[m[i].aux]++;
        int left = cnt[0] + cnt[1], right = cnt[2] + cnt[3];
        int valid = (left < right ? left : right) + cnt[4];
        num = valid < 80 ? valid : 80;
    } else num = n < 80 ? n : 80;
    return (double)num + sds * ((double)num / 8) + pds * ((double)num / 8);
}

/* consolidate one cluster (consolidate_clusters_unilocal :214-228 / _bilocal :231-303) */
static void consolidate(svo_ctx* c, const csig* m, const int32_t* midx, int n, int64_t part) {
    svx_cluster_view* v = &c->clu;
    clu_reserve(c, v->n + 1, v->n_members + n);
    int64_t k = v->n++;
    int type = m[0].type;
    double xs[128], xp[128];
    int64_t ss = 0, se = 0;
    for (int i = 0; i < n; i++) { ss += m[i].start; se += m[i].end; }
    double avg_s = (double)ss / (double)n, avg_e = (double)se / (double)n;
    int has = n > 1;
    double std_span = NAN, std_pos = NAN;
    if (has) {
        for (int i = 0; i < n; i++) { xs[i] = (double)(m[i].end - m[i].start); xp[i] = (double)(m[i].end + m[i].start) / 2.0; }
        std_span = stdev_d(xs, n); std_pos = stdev_d(xp, n);
    }
    v->type[k] = (uint8_t)type; v->contig[k] = m[0].contig; v->size[k] = n; v->aux[k] = 0;
    v->start[k] = (int32_t)py_round_half_even(avg_s); v->end[k] = (int32_t)py_round_half_even(avg_e);
    v->contig2[k] = -1; v->start2[k] = 0; v->end2[k] = 0;
    if (type == SVX_DEL || type == SVX_INS || type == SVX_INV) {
        v->score[k] = calc_score(m, n, has, std_span, std_pos, avg_e - avg_s, type);
        v->std_span[k] = std_span; v->std_pos[k] = std_pos;
    } else if (type == SVX_DUP_TAN) {
        int64_t maxc = 0; for (int i = 0; i < n; i++) if (m[i].pos2 > maxc) maxc = m[i].pos2;
        v->score[k] = calc_score(m, n, has, std_span, std_pos, avg_e - avg_s, type);
        v->contig2[k] = m[0].contig; v->start2[k] = v->end[k];
        v->end2[k] = (int32_t)(v->end[k] + maxc * ((int64_t)v->end[k] - v->start[k]));
        v->std_span[k] = std_span; v->std_pos[k] = std_pos;
    } else if (type == SVX_DUP_INT) {
        int64_t ds = 0, de = 0;
        for (int i = 0; i < n; i++) { ds += m[i].pos2; de += m[i].pos2 + (m[i].end - m[i].start); }
        double davg_s = (double)ds / (double)n, davg_e = (double)de / (double)n;
        double span = ((avg_e - avg_s) + (davg_e - davg_s)) / 2.0;       /* mean([a, b]) */
        v->contig2[k] = m[0].contig2; v->start2[k] = (int32_t)py_round_half_even(davg_s); v->end2[k] = (int32_t)py_round_half_even(davg_e);
        if (has) {
            double dsp, dpo;
            for (int i = 0; i < n; i++) { xs[i] = (double)(m[i].end - m[i].start); xp[i] = (double)(m[i].pos2 + (m[i].end - m[i].start) + m[i].pos2) / 2.0; }
            dsp = stdev_d(xs, n); dpo = stdev_d(xp, n);
            double mspan = (std_span + dsp) / 2.0, mpos = (std_pos + dpo) / 2.0;
            v->score[k] = calc_score(m, n, 1, mspan, mpos, span, type);
            v->std_span[k] = mspan; v->std_pos[k] = mpos;
        } else { v->score[k] = calc_score(m, n, 0, 0, 0, span, type); v->std_span[k] = NAN; v->std_pos[k] = NAN; }
    } else { /* BND */
        int64_t ds = 0; for (int i = 0; i < n; i++) ds += m[i].pos2;
        double davg_s = (double)ds / (double)n, davg_e = (double)(ds + n) / (double)n;
        v->contig2[k] = m[0].contig2; v->start2[k] = (int32_t)py_round_half_even(davg_s); v->end2[k] = (int32_t)py_round_half_even(davg_e);
        v->aux[k] = (uint8_t)m[0].aux;
        if (has) {
            for (int i = 0; i < n; i++) xp[i] = (double)(m[i].pos2 + 1 + m[i].pos2) / 2.0;
            double dpo = stdev_d(xp, n);
            v->score[k] = calc_score(m, n, 1, std_pos, dpo, 500.0, type);
            v->std_span[k] = std_pos; v->std_pos[k] = dpo;
        } else { v->score[k] = calc_score(m, n, 0, 0, 0, 500.0, type); v->std_span[k] = NAN; v->std_pos[k] = NAN; }
    }
    v->member_off[k] = v->n_members;
    for (int i = 0; i < n; i++) v->members[v->n_members++] = midx[i];
    v->member_off[k + 1] = v->n_members;
    c->part_index[k] = part;
}

/* cluster_sv_signatures -> partition_and_cluster x6 (src/svim/SVIM_CLUSTER.py:7-26, SVIM_clustering.py:375-385,
 * form_partitions :17-29, clusters_from_partitions :122-180) */
/* one partition's sample (pass 1), its same-read duplicates removed and its condensed distance matrix (pass 2) */
typedef struct {
    int type, ns, nm;
    int32_t midx[100];              /* sampled members: indices into the signature table, sample order */
    int32_t mmidx[100];             /* ... without the same-read duplicates */
    double* cd;                     /* nm (nm - 1) / 2 distances, scipy's condensed order */
} part_job;
typedef struct { svo_ctx* c; const svx_sig_view* v; const svx_params* p; part_job* jobs; int64_t n; int64_t next; } part_work;

static void part_distances(svo_ctx* c, const svx_sig_view* v, const svx_params* p, part_job* J) {
    const int ns = J->ns, type = J->type;
    csig m[100];
    for (int k = 0; k < ns; k++) m[k] = get_sig(v, J->midx[k]);
    /* same-read duplicate removal (:141-151); INV exempt */
    unsigned char dup[100]; memset(dup, 0, sizeof dup);
    if (type != SVX_INV) {
        for (int i = 0; i < ns - 1; i++) for (int j = i + 1; j < ns; j++)
            if (m[i].read_id == m[j].read_id) {
                double d = span_position_distance(c, &m[i], &m[j], p);
                if (d <= p->cluster_max_distance) dup[j] = 1;
            }
    }
    csig mm[100]; int nm = 0;
    for (int k = 0; k < ns; k++) if (!dup[k]) { mm[nm] = m[k]; J->mmidx[nm] = J->midx[k]; nm++; }
    J->nm = nm; J->cd = NULL;
    if (nm > 1) {
        double* cd = malloc(sizeof(double) * (size_t)(nm * (nm - 1) / 2));
        int64_t q = 0;
        for (int i = 0; i < nm - 1; i++) for (int j = i + 1; j < nm; j++) {
            if (type != SVX_INV && mm[i].read_id == mm[j].read_id) cd[q++] = 99999.0;
            else cd[q++] = span_position_distance(c, &mm[i], &mm[j], p);
        }
        J->cd = cd;
    }
}
static void* part_worker(void* arg) {
    part_work* W = arg;
    for (;;) {
        const int64_t i = __atomic_fetch_add(&W->next, 1, __ATOMIC_RELAXED);
        if (i >= W->n) break;
        part_distances(W->c, W->v, W->p, &W->jobs[i]);
    }
    return NULL;
}
/* test infrastructure only: the pair distances of svo_cluster on n worker threads (results do not depend on n) */
int svo_set_threads(svo_ctx* c, int n) { c->n_threads = n < 1 ? 1 : n; return 0; }

int svo_cluster(svo_ctx* c, int source, const svx_sig_view* sigs_in, int32_t n_contig, const int32_t* rank, const svx_params* p) {
    (void)n_contig;
    svx_sig_view local; const svx_sig_view* v = sigs_in;
    if (source == 0 || source == 1) {
        sigtab* t = source ? &c->bnd : &c->sig;
        memset(&local, 0, sizeof local);
        local.n = t->n; local.key = t->key; local.type = t->type; local.src = t->src; local.aux = t->aux; local.contig = t->contig;
        local.start = t->start; local.end = t->end; local.contig2 = t->contig2; local.pos2 = t->pos2; local.read_id = t->read_id;
        if (source == 0) { local.seq_off = c->sig_seq_off; local.seq = c->sig_seq; }
        v = &local;
    }
    int64_t n = v->n;
    svx_cluster_view* out = &c->clu;
    out->n = 0; out->n_members = 0; memset(out->type_count, 0, sizeof out->type_count);
    clu_reserve(c, 1, 1); out->member_off[0] = 0;
    int64_t e_pairs0 = c->stats.n_edit_pairs; (void)e_pairs0;
    c->stats.n_partitions = c->stats.n_large_partitions = c->stats.n_pairs = 0; c->stats.n_edit_pairs = c->stats.n_edit_cells = 0;
    skey* keys = malloc(sizeof(skey) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        skey k; k.type = v->type[i]; k.idx = i; k.r2 = 0;
        switch (k.type) {
            case SVX_INS: k.r1 = rank[v->contig[i]]; k.coord = v->start[i]; break;
            case SVX_DUP_INT: k.r1 = rank[v->contig2[i]]; k.r2 = rank[v->contig[i]]; k.coord = v->pos2[i]; break;
            case SVX_BND: k.r1 = rank[v->contig[i]]; k.coord = v->start[i]; break;
            default: k.r1 = rank[v->contig[i]]; k.coord = v->end[i]; break;
        }
        keys[i] = k;
    }
    qsort(keys, (size_t)n, sizeof(skey), cmp_skey);
    int64_t global_part = 0;
    int64_t i0 = 0;
    int64_t chain_words[SVX_NTYPES] = {0, 0, 0, 0, 0, 0}, chain_end[SVX_NTYPES];
    if (c->chain_fn && c->chain_fn(c->chain_user, 0, chain_words) != 0) return -5;
    for (int type = 0; type < SVX_NTYPES; type++) {
        int64_t t_begin = i0;
        while (i0 < n && keys[i0].type == type) i0++;
        int64_t t_end = i0;
        int64_t first_cluster = out->n;
        mt rng; mt_seed_int(&rng, 1524u);                     /* seed(1524) once per type call, :129 */
        for (int64_t w = 0; w < chain_words[type]; w++) (void)mt_u32(&rng);     /* svo_cluster_set_chain: words earlier ranks consumed */
        /* Three passes over the type's partitions.  (1) boundaries + random.sample, in order (the generator is one stream per type);
         * (2) the pair distances of every partition - independent of each other, the only expensive part (edit distances): spread over
         * svo_set_threads worker threads, results the same whatever the thread count; (3) linkage + consolidation, in order. */
        int64_t np_cap = 1024, np = 0;
        part_job* jobs = malloc(sizeof(part_job) * (size_t)np_cap);
        int64_t ps = t_begin;
        while (ps < t_end) {
            /* grow the partition while downstream_distance_to(prev, cur) <= max_distance */
            int64_t pe = ps + 1;
            while (pe < t_end) {
                int64_t a = keys[pe - 1].idx, b = keys[pe].idx;
                int64_t dist; int inf = 0;
                if (type == SVX_INS) { if (v->contig[a] != v->contig[b]) inf = 1; dist = (int64_t)v->start[b] - v->start[a]; }
                else if (type == SVX_DUP_INT) { if (v->contig2[a] != v->contig2[b] || v->contig[a] != v->contig[b]) inf = 1; dist = (int64_t)v->pos2[b] - v->pos2[a]; }
                else { if (v->contig[a] != v->contig[b]) inf = 1; dist = (int64_t)v->start[b] - v->end[a]; }
                if (dist < 0) dist = 0;
                if (inf || dist > p->partition_max_distance) break;
                pe++;
            }
            int64_t psize = pe - ps;
            if (np == np_cap) { np_cap *= 2; jobs = realloc(jobs, sizeof(part_job) * (size_t)np_cap); }
            part_job* J = &jobs[np++];
            memset(J, 0, sizeof *J);
            c->stats.n_partitions++;
            int32_t sample[100];
            if (psize > 100) { mt_sample100(&rng, psize, sample); J->ns = 100; c->stats.n_large_partitions++; }
            else { J->ns = (int)psize; for (int k = 0; k < J->ns; k++) sample[k] = k; }
            for (int k = 0; k < J->ns; k++) J->midx[k] = (int32_t)keys[ps + sample[k]].idx;
            J->type = type;
            ps = pe;
        }
        {
            part_work W; W.c = c; W.v = v; W.p = p; W.jobs = jobs; W.n = np; W.next = 0;
            int nt = c->n_threads > 1 ? c->n_threads : 1;
            if (nt > np) nt = (int)(np ? np : 1);
            if (nt <= 1) part_worker(&W);
            else {
                pthread_t* th = malloc(sizeof(pthread_t) * (size_t)nt);
                int started = 0;
                for (int k = 0; k < nt - 1; k++) if (pthread_create(&th[started], NULL, part_worker, &W) == 0) started++;
                part_worker(&W);
                for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
                free(th);
            }
        }
        for (int64_t pi = 0; pi < np; pi++) {
            part_job* J = &jobs[pi];
            const int nm = J->nm;
            csig mm[100];
            for (int k = 0; k < nm; k++) mm[k] = get_sig(v, J->mmidx[k]);
            if (nm == 1) consolidate(c, mm, J->mmidx, 1, global_part);
            else {
                c->stats.n_pairs += (int64_t)nm * (nm - 1) / 2;
                int32_t lab[100];
                svo_linkage_fcluster(nm, J->cd, p->cluster_max_distance, lab, NULL);
                int maxl = 0; for (int k = 0; k < nm; k++) if (lab[k] > maxl) maxl = lab[k];
                for (int l = 1; l <= maxl; l++) {
                    csig cm[100]; int32_t cidx_[100]; int cn = 0;
                    for (int k = 0; k < nm; k++) if (lab[k] == l) { cm[cn] = mm[k]; cidx_[cn] = J->mmidx[k]; cn++; }
                    if (cn) consolidate(c, cm, cidx_, cn, global_part);
                }
                free(J->cd);
            }
            global_part++;
        }
        free(jobs);
        /* unilocal types: sorted(key=(contig, (end+start)/2)) - stable (:381) */
        int64_t ncl = out->n - first_cluster;
        if (type <= SVX_INV && ncl > 1) {
            skey* ck = malloc(sizeof(skey) * (size_t)ncl);
            for (int64_t k = 0; k < ncl; k++) { int64_t g = first_cluster + k; ck[k].type = 0; ck[k].r1 = rank[out->contig[g]]; ck[k].r2 = 0;
                ck[k].coord = (int64_t)out->start[g] + out->end[g]; ck[k].idx = k; }
            qsort(ck, (size_t)ncl, sizeof(skey), cmp_skey);
            /* permute cluster records + member lists */
            svx_cluster_view tmp; memset(&tmp, 0, sizeof tmp);
#define CP(f) tmp.f = malloc((size_t)ncl * sizeof(*tmp.f)); for (int64_t k = 0; k < ncl; k++) tmp.f[k] = out->f[first_cluster + ck[k].idx]; \
            memcpy(out->f + first_cluster, tmp.f, (size_t)ncl * sizeof(*tmp.f)); free(tmp.f)
            int64_t mbase = out->member_off[first_cluster], mtot = out->n_members - mbase;
            int32_t* newm = malloc((size_t)(mtot ? mtot : 1) * 4); int64_t* newoff = malloc((size_t)(ncl + 1) * 8);
            int64_t w = 0;
            for (int64_t k = 0; k < ncl; k++) { int64_t g = first_cluster + ck[k].idx; newoff[k] = mbase + w;
                for (int64_t q = out->member_off[g]; q < out->member_off[g + 1]; q++) newm[w++] = out->members[q]; }
            newoff[ncl] = mbase + w;
            int64_t* pi = malloc((size_t)ncl * 8); for (int64_t k = 0; k < ncl; k++) pi[k] = c->part_index[first_cluster + ck[k].idx];
            memcpy(c->part_index + first_cluster, pi, (size_t)ncl * 8); free(pi);
            CP(type); CP(contig); CP(start); CP(end); CP(contig2); CP(start2); CP(end2); CP(aux); CP(score); CP(std_span); CP(std_pos); CP(size);
#undef CP
            memcpy(out->members + mbase, newm, (size_t)mtot * 4);
            memcpy(out->member_off + first_cluster, newoff, (size_t)(ncl + 1) * 8);
            free(newm); free(newoff); free(ck);
        }
        out->type_count[type] = ncl;
        chain_end[type] = rng.drawn;
    }
    free(keys);
    c->stats.n_clusters = out->n;
    if (c->chain_fn && c->chain_fn(c->chain_user, 1, chain_end) != 0) return -5;
    return 0;
}
int svo_cluster_set_chain(svo_ctx* c, svo_chain_fn fn, void* user) { c->chain_fn = fn; c->chain_user = user; return 0; }
int svo_cluster_count(svo_ctx* c, int64_t* ncl, int64_t* nmem) { *ncl = c->clu.n; *nmem = c->clu.n_members; return 0; }
int svo_cluster_fetch(svo_ctx* c, svx_cluster_view* o) {
    svx_cluster_view* v = &c->clu; size_t n = (size_t)v->n;
    memcpy(o->type, v->type, n); memcpy(o->contig, v->contig, n * 4); memcpy(o->start, v->start, n * 4); memcpy(o->end, v->end, n * 4);
    memcpy(o->contig2, v->contig2, n * 4); memcpy(o->start2, v->start2, n * 4); memcpy(o->end2, v->end2, n * 4); memcpy(o->aux, v->aux, n);
    memcpy(o->score, v->score, n * 8); memcpy(o->std_span, v->std_span, n * 8); memcpy(o->std_pos, v->std_pos, n * 8);
    memcpy(o->size, v->size, n * 4); memcpy(o->member_off, v->member_off, (n + 1) * 8); memcpy(o->members, v->members, (size_t)v->n_members * 4);
    o->n = v->n; o->n_members = v->n_members; memcpy(o->type_count, v->type_count, sizeof v->type_count);
    return 0;
}
int svo_cluster_fetch_part_index(svo_ctx* c, int64_t* out) { memcpy(out, c->part_index, (size_t)c->clu.n * 8); return 0; }

/* form_partitions alone: partition id per signature in sorted order (for G4) */
int svo_form_partitions(const svx_sig_view* v, const int32_t* rank, int64_t max_distance, int64_t* sorted_idx, int64_t* part_id) {
    int64_t n = v->n;
    skey* keys = malloc(sizeof(skey) * (size_t)(n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        skey k; k.type = v->type[i]; k.idx = i; k.r2 = 0;
        switch (k.type) {
            case SVX_INS: k.r1 = rank[v->contig[i]]; k.coord = v->start[i]; break;
            case SVX_DUP_INT: k.r1 = rank[v->contig2[i]]; k.r2 = rank[v->contig[i]]; k.coord = v->pos2[i]; break;
            case SVX_BND: k.r1 = rank[v->contig[i]]; k.coord = v->start[i]; break;
            default: k.r1 = rank[v->contig[i]]; k.coord = v->end[i]; break;
        }
        keys[i] = k;
    }
    qsort(keys, (size_t)n, sizeof(skey), cmp_skey);
    int64_t pid = -1;
    for (int64_t i = 0; i < n; i++) {
        int newp = 1;
        if (i > 0 && keys[i - 1].type == keys[i].type) {
            int64_t a = keys[i - 1].idx, b = keys[i].idx; int type = keys[i].type; int64_t dist; int inf = 0;
            if (type == SVX_INS) { if (v->contig[a] != v->contig[b]) inf = 1; dist = (int64_t)v->start[b] - v->start[a]; }
            else if (type == SVX_DUP_INT) { if (v->contig2[a] != v->contig2[b] || v->contig[a] != v->contig[b]) inf = 1; dist = (int64_t)v->pos2[b] - v->pos2[a]; }
            else { if (v->contig[a] != v->contig[b]) inf = 1; dist = (int64_t)v->start[b] - v->end[a]; }
            if (dist < 0) dist = 0;
            newp = inf || dist > max_distance;
        }
        if (newp) pid++;
        sorted_idx[i] = keys[i].idx; part_id[i] = pid;
    }
    free(keys);
    return 0;
}

/* ---------------------------------------------------------------- GENOTYPE ----
 * genotype() of src/svim/SVIM_genotyping.py:34-93, the part that touches alignments: for every candidate walk
 * bam.fetch(contig, max(0, start-1000), min(contig_length, end+1000)) (:47-48) in file order, skip reads of the variant (:63) and
 * unmapped / secondary / low-mapq alignments (:65), stop after 500 eligible ones (:56,68) and collect the names of those spanning
 * the locus (:70-77).  fetch() = htslib's overlap rule over a coordinate-sorted file: pos < stop and bam_endpos > start, where
 * bam_endpos is pos + 1 for a record without reference span. */
static svx_aln_index g_index;

int svo_set_alignment_index(svo_ctx* c, const svx_aln_index* h) { (void)c; g_index = *h; return 0; }   /* borrows the caller's arrays */

int svo_genotype(svo_ctx* c, int32_t mode, int64_t n_cand, const int32_t* cand_tid, const int32_t* cand_start, const int32_t* cand_end,
                 const int64_t* member_off, const int32_t* member_names, int32_t min_mapq, int32_t* out_ref_reads) {
    (void)c;
    const svx_aln_index* ix = &g_index;
    int32_t names[500];
    for (int64_t k = 0; k < n_cand; k++) {
        out_ref_reads[k] = 0;
        const int tid = cand_tid[k];
        if (tid < 0 || tid >= ix->n_contig) continue;
        const int64_t start = cand_start[k], end = cand_end[k];
        const int64_t clen = ix->contig_len[tid];
        const int64_t ws = start - 1000 > 0 ? start - 1000 : 0, we = end + 1000 < clen ? end + 1000 : clen;
        const double minimum_overlap = fmin((double)(end - start) / 2.0, 2000.0);          /* :71 */
        int aln_no = 0, n_names = 0;
        for (int64_t i = ix->contig_first[tid]; i < ix->contig_first[tid + 1] && aln_no < 500; i++) {
            const int64_t rs = ix->pos[i], re = ix->end[i];
            const int64_t endp = re > rs ? re : rs + 1;
            if (!(rs < we && endp > ws)) continue;                                          /* not returned by fetch */
            const int32_t name = ix->name_id[i];
            int in_variant = 0;
            for (int64_t m = member_off[k]; m < member_off[k + 1]; m++) if (member_names[m] == name) { in_variant = 1; break; }
            if (in_variant) continue;                                                       /* :63 */
            if ((ix->flag[i] & 0x4) || (ix->flag[i] & 0x100) || ix->mapq[i] < min_mapq) continue;   /* :65 */
            aln_no++;                                                                       /* :68 */
            int support;
            if (mode == 0)
                support = ((double)rs < (double)end - minimum_overlap && re > end + 100) || (rs < start - 100 && (double)re > (double)start + minimum_overlap);
            else
                support = rs < start - 100 && re > end + 100;
            if (!support) continue;
            int seen = 0;
            for (int j = 0; j < n_names; j++) if (names[j] == name) { seen = 1; break; }
            if (!seen) names[n_names++] = name;
        }
        out_ref_reads[k] = n_names;
    }
    return 0;
}
