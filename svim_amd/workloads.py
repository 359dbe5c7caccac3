"""Seeded stand-ins for BASELINE.json's configurations, built with torch ON THE DEVICE (bench.py, large parity tests).

    c1  configs[1]  1 M ONT reads (N50 20 kb), one 250 Mb contig, DEL/INS in the CIGAR + INV split reads   -> devsynth.make_batch
    c2  configs[2]  PacBio-HiFi profile: 15-20 kb reads, sparse CIGAR noise (10^2-10^3 ops per read), three contigs
                    (chr1, chr10, chr2 - the str order of the names differs from the header order), the FULL SV-type set:
                    CIGAR DEL/INS plus split reads that make the reference emit DEL, INS (sequence taken from the primary:
                    the edlib path), INV, DUP_TAN, BND (other contig and > max_sv_size) and DUP_INT
                    (src/svim/SVIM_inter.py:58-300)
    c3  configs[3]  whole-genome ONT profile: 47 contigs with hg38's names and length ratios (header order chr1..chr22, X, Y, M, then alt / random /
                    unplaced scaffolds and HLA alleles - their str order, which the reference sorts partitions by, interleaves them), every SV type,
                    split reads across contigs: what a contig-sharded multi-GPU run sees (tests: 8 ranks over one file, tests/mp_c3_ranks_one_gpu.py)
    c4  configs[4]  60x PacBio-CLR profile: shorter, noisier reads and DENSE sites, so that --partition_max_distance in
                    {1000, 5000, 20000, 100000} produces many partitions beyond 100 and beyond 1045 signatures
                    (random.sample pool / set paths, src/svim/SVIM_clustering.py:132-134)

The real inputs of configs[2]-[4] are BAM files that do not exist in this environment; `bench.py --bam --fasta` runs one when
it is there.  What the generators plant is decided by the oracle and the GPU path alike - parity never depends on the
generator being "right", only the type mix reported beside the rates does.

make_batch_full generalises devsynth.make_batch: several contigs, and per read one of eight layouts.  A split read's primary
is forward and ENDS exactly at its site's (jittered) breakpoint A; the rest of the read is its trailing soft clip, described by
one or two SA-derived segment rows (and the matching supplementary records with S/M/S CIGARs):

    kind            segments after the primary [.., A)                              reference emits
    1 INV           rev [A, A+size) ; fwd [A+size, ..)                              INV left_fwd + INV right_fwd
    2 split DEL     fwd [A+size, ..)                                                DEL(A, A+size)
    3 split INS     fwd [A, ..) after a gap of `size` read bases                    INS(A, A+size) + inserted bases from the primary
    4 DUP_TAN       fwd [A-size, ..)                                                DUP_TAN(A-size, A)
    5 BND (contig)  fwd on the partner contig                                       BND
    6 BND (far)     fwd [A + max_sv_size + 1000 + size, ..)                         BND
    7 DUP_INT       fwd [S, S+size) far away / other contig ; fwd [A, ..)           2 BND + DUP_INT(S, S+size -> A)
"""
import math

import torch

from .devsynth import DeviceBatch, _lognormal_lengths, _rand_acgt_codes, _rand_acgt_pairs

KIND_NAMES = ("plain", "INV", "split_DEL", "split_INS", "DUP_TAN", "BND_contig", "BND_far", "DUP_INT")

PROFILES = {
    # HiFi: accurate reads -> long match runs; every split layout
    "c2": dict(contigs=(("chr1", 120_000_000), ("chr10", 70_000_000), ("chr2", 60_000_000)), reads_per_mb=1715, length="normal",
               len_mean=17500, len_sd=2000, m_lo=20, m_hi=110, ind_hi=2, sites_per_mb=100, size_lo=50, size_hi=6000,
               site_mix=(0.30, 0.30, 0.06, 0.08, 0.06, 0.08, 0.04, 0.03, 0.05), ins_err=0.005, max_sites_per_read=6, lowq=0.03),
    # CLR 60x: noisy, dense sites (one per ~1.5 kb: neighbouring sites chain into large partitions)
    "c4": dict(contigs=(("chr1", 40_000_000),), reads_per_mb=5000, length="lognormal", n50=14000, m_lo=3, m_hi=14, ind_hi=3,
               sites_per_mb=650, size_lo=50, size_hi=2500, site_mix=(0.47, 0.47, 0.03, 0.0, 0.0, 0.03, 0.0, 0.0, 0.0), ins_err=0.08,
               max_sites_per_read=16, lowq=0.03),
    # configs[3] stand-in: a whole genome's worth of contigs in header order chr1..chr22, X, Y, M followed by alt / random / unplaced scaffolds (hg38 names
    # and length RATIOS; the str order of the names - the reference's partition key - interleaves them: chr1, chr10, chr11, .., chr1_KI270706v1_random,
    # chr2, ..), ONT-like reads, every split-read layout: BND and DUP_INT rows cross contigs, i.e. ranks
    "c3": dict(contigs=tuple([("chr%s" % n, l) for n, l in zip(list(range(1, 23)) + ["X", "Y", "M"],
                              (248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
                               133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
                               156040895, 57227415, 16569))] +
                             [(n, 1) for n in ("chr1_KI270706v1_random", "chr1_KI270707v1_random", "chr2_KI270715v1_random", "chr4_GL000008v2_random",
                                               "chr9_KI270717v1_random", "chr11_KI270721v1_random", "chr14_GL000009v2_random", "chr14_GL000194v1_random",
                                               "chr17_GL000205v2_random", "chr22_KI270731v1_random", "chrUn_KI270302v1", "chrUn_GL000195v1", "chrUn_KI270442v1",
                                               "chrUn_GL000214v1", "chrUn_KI270744v1", "chrEBV", "chr1_KI270762v1_alt", "chr6_GL000250v2_alt",
                                               "chr17_KI270857v1_alt", "chr19_KI270938v1_alt", "HLA-A*01:01:01:01", "HLA-DRB1*15:03:01:01")]),
               reads_per_mb=1500, length="lognormal", n50=20000, m_lo=5, m_hi=30, ind_hi=3, sites_per_mb=120, size_lo=50, size_hi=5000,
               site_mix=(0.33, 0.33, 0.08, 0.05, 0.05, 0.04, 0.06, 0.02, 0.04), ins_err=0.03, max_sites_per_read=6, lowq=0.03),
}
# site_mix: fractions of (CIGAR DEL, CIGAR INS, INV, split DEL, split INS, DUP_TAN, BND contig, BND far, DUP_INT)
_SITE_KIND = (0, 0, 1, 2, 3, 4, 5, 6, 7)          # split-read layout a site of that class produces (0 = carried in the CIGAR)


def profile(name, scale=1.0):
    """A named profile with contig lengths (hence read and site counts) scaled by `scale`."""
    p = dict(PROFILES[name])
    p["contigs"] = tuple((n, max(400_000, int(l * scale))) for n, l in p["contigs"])
    p["name"] = name
    return p


def make_batch_full(prof, seed=3, device="cuda", split_read_frac=0.10, max_sv_size=100000):
    """-> (DeviceBatch, genome codes uint8 [sum of contig lengths], genome offsets int64 [n_contig+1], meta dict)"""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    i64, i32 = torch.int64, torch.int32
    names = [c[0] for c in prof["contigs"]]
    clen = torch.tensor([c[1] for c in prof["contigs"]], dtype=i64, device=dev)
    NC = len(names)
    coff = torch.zeros(NC + 1, dtype=i64, device=dev)
    torch.cumsum(clen, 0, out=coff[1:])
    total = int(coff[-1].item())
    R = max(64, int(prof["reads_per_mb"] * total / 1e6))
    S = max(16, int(prof["sites_per_mb"] * total / 1e6))
    size_lo, size_hi = prof["size_lo"], prof["size_hi"]
    margin = 260_000 if total > 4_000_000 else 130_000                     # room for the longest read + far-BND partner inside a contig

    def rnd(n):
        return torch.rand(n, generator=gen, device=dev)

    def rint(lo, hi, n):
        return torch.randint(lo, hi, (n,), generator=gen, device=dev, dtype=i64)

    def place(n, lo_margin, hi_margin):
        """uniform positions in GLOBAL coordinates (contig offset + pos), each inside one contig with the given margins"""
        c = torch.searchsorted(coff[1:], (rnd(n) * total).to(i64), right=True).clamp_max(NC - 1)
        span = (clen[c] - lo_margin - hi_margin).clamp_min(1)
        return c, coff[c] + lo_margin + (rnd(n) * span.double()).to(i64)

    # ---- sites (global coordinates, sorted) ---------------------------------------------------------------------------
    s_contig, s_gpos = place(S, 20000, margin)
    order = torch.argsort(s_gpos)
    s_contig, s_gpos = s_contig[order], s_gpos[order]
    mix = torch.tensor(prof["site_mix"], dtype=torch.float64, device=dev)
    mix = torch.cumsum(mix / mix.sum(), 0)
    s_class = torch.searchsorted(mix, rnd(S).double()).clamp_max(8)       # 0..8, see site_mix
    s_kind = torch.tensor(_SITE_KIND, dtype=i64, device=dev)[s_class]
    s_size = torch.exp(math.log(size_lo) + rnd(S) * math.log(size_hi / size_lo)).to(i64)
    s_hom = rnd(S) < 0.5
    if NC == 1:                                                            # no partner contig: those sites become far BNDs
        s_kind = torch.where(s_kind == 5, torch.full_like(s_kind, 6), s_kind)
    # partner locus of BND (contig) / source locus of DUP_INT: another contig when there is one (DUP_INT: half of them far on the same)
    p_contig = (s_contig + 1 + (rint(0, max(1, NC - 1), S) if NC > 1 else 0)) % NC
    same = (s_kind == 7) & ((rnd(S) < 0.5) | (NC == 1))
    p_contig = torch.where(same, s_contig, p_contig)
    p_pos = 20000 + (rnd(S) * (clen[p_contig] - 20000 - margin).clamp_min(1).double()).to(i64)
    s_pos = s_gpos - coff[s_contig]
    far = torch.where(s_pos + max_sv_size + 30000 + size_hi < clen[s_contig] - 40000, s_pos + max_sv_size + 20000, s_pos - max_sv_size - 20000 - s_size)
    p_pos = torch.where(same, far.clamp_min(1000), p_pos)
    max_ins = int(size_hi) + 8
    seq_sites = torch.nonzero((s_class == 1) | (s_class == 4)).flatten()   # INS sites (CIGAR and split) own an inserted sequence
    seq_slot = torch.full((S,), -1, dtype=i64, device=dev)
    seq_slot[seq_sites] = torch.arange(seq_sites.numel(), device=dev)
    site_seq = (1 << torch.randint(0, 4, (max(1, seq_sites.numel()), max_ins), generator=gen, device=dev, dtype=i64)).to(torch.uint8)
    split_sites = torch.nonzero(s_kind > 0).flatten()

    # ---- reads -----------------------------------------------------------------------------------------------------------
    if prof["length"] == "normal":
        L = (prof["len_mean"] + prof["len_sd"] * torch.randn(R, generator=gen, device=dev)).clamp_(3000, 40000).to(i64)
    else:
        L = _lognormal_lengths(gen, R, prof["n50"], dev, lo=1000, hi=120000)
    r_contig, r_gstart = place(R, 1000, margin)
    is_split = (rnd(R) < split_read_frac) & (split_sites.numel() > 0)
    pick = split_sites[rint(0, max(1, split_sites.numel()), R)] if split_sites.numel() else torch.zeros(R, dtype=i64, device=dev)
    kind = torch.where(is_split, s_kind[pick], torch.zeros(R, dtype=i64, device=dev))
    is_split = kind > 0
    bp_a = s_pos[pick] + rint(-5, 6, R)                                   # breakpoint A (contig coordinates) of a split read
    size = (s_size[pick] + rint(-5, 6, R)).clamp_min(40)
    r_contig = torch.where(is_split, s_contig[pick], r_contig)
    hap = rnd(R) < 0.5
    lead = torch.where(rnd(R) < 0.5, rint(1, 60, R), torch.zeros(R, dtype=i64, device=dev))
    # ---- CIGAR units: (M run, short indel) ----------------------------------------------------------------------------
    m_lo, m_hi, ind_hi = prof["m_lo"], prof["m_hi"], prof["ind_hi"]
    unit_mean = (m_lo + m_hi) / 2.0 + (1 + ind_hi) / 4.0
    n_units = (L.double() / unit_mean).to(i64).clamp_min(2)
    uo = torch.zeros(R + 1, dtype=i64, device=dev)
    torch.cumsum(n_units, 0, out=uo[1:])
    U = int(uo[-1].item())
    unit_read = torch.repeat_interleave(torch.arange(R, device=dev, dtype=i32), n_units)
    m_len = torch.randint(m_lo, m_hi + 1, (U,), generator=gen, device=dev, dtype=i32)
    ind_len = torch.randint(1, ind_hi + 1, (U,), generator=gen, device=dev, dtype=i32)
    ind_op = torch.randint(1, 3, (U,), generator=gen, device=dev, dtype=torch.int8)      # 1 = I, 2 = D
    last_unit = uo[1:] - 1
    ind_op[last_unit] = 0                                                                   # a CIGAR core must not end in I/D
    zero1 = torch.zeros(1, dtype=i64, device=dev)
    cum = torch.cumsum(m_len.to(i64) + torch.where(ind_op != 1, ind_len, 0).to(i64), 0)
    r_noise = cum[last_unit] - torch.cat([zero1, cum])[uo[:-1]]
    del cum
    start = r_gstart - coff[r_contig]                                       # contig coordinates
    start = torch.where(is_split, (bp_a - r_noise).clamp_min(100), start)
    gstart = coff[r_contig] + start
    # ---- plant CIGAR DEL / INS sites, one per pass ------------------------------------------------------------------------
    planted_unit, planted_site = [], []
    for j in range(prof["max_sites_per_read"]):
        cum = torch.cumsum(m_len.to(i64) + torch.where(ind_op != 1, ind_len, 0).to(i64), 0)
        base = torch.where(uo[:-1] > 0, cum[(uo[:-1] - 1).clamp_min(0)], torch.zeros_like(uo[:-1]))
        read_ref_end = gstart + (cum[last_unit] - base)
        lo = torch.searchsorted(s_gpos, gstart + 300)
        hi = torch.searchsorted(s_gpos, read_ref_end - 300 - size_hi)
        sidx = lo + j
        ok = sidx < hi
        sidx = sidx.clamp_max(S - 1)
        ok &= (s_kind[sidx] == 0) & (s_hom[sidx] | hap) & ~is_split & (s_contig[sidx] == r_contig)
        target = s_gpos[sidx] - gstart + base
        after_m = cum - torch.where(ind_op != 1, ind_len, 0).to(i64)
        uidx = torch.searchsorted(after_m, target)
        ok &= (uidx > uo[:-1]) & (uidx < last_unit - 1)
        rsel = torch.nonzero(ok).flatten()
        if rsel.numel() == 0:
            continue
        us, ss = uidx[rsel], sidx[rsel]
        jit = rint(-5, 6, rsel.numel())
        ind_len[us] = (s_size[ss] + jit).clamp_min(1).to(i32)
        ind_op[us] = torch.where(s_class[ss] == 0, 2, 1).to(torch.int8)
        planted_unit.append(us)
        planted_site.append(ss)
    planted_unit = torch.cat(planted_unit) if planted_unit else torch.zeros(0, dtype=i64, device=dev)
    planted_site = torch.cat(planted_site) if planted_site else torch.zeros(0, dtype=i64, device=dev)
    if planted_unit.numel():
        # a unit planted in two passes keeps the LAST site (see devsynth.make_batch)
        ordk = torch.arange(planted_unit.numel(), device=dev)
        srt = torch.argsort(planted_unit * (planted_unit.numel() + 1) + ordk)
        pu_s = planted_unit[srt]
        last = torch.ones_like(pu_s, dtype=torch.bool)
        last[:-1] = pu_s[:-1] != pu_s[1:]
        keep = srt[last]
        planted_unit, planted_site = planted_unit[keep], planted_site[keep]
    # ---- per-read totals ----------------------------------------------------------------------------------------------------
    ref_c = m_len.to(i64) + torch.where(ind_op != 1, ind_len, 0).to(i64)
    qry_c = m_len.to(i64) + torch.where(ind_op != 2, ind_len, 0).to(i64)
    cum_q = torch.cumsum(qry_c, 0)
    cum_r = torch.cumsum(ref_c, 0)
    q_base = torch.cat([zero1, cum_q])[uo[:-1]]
    r_base = torch.cat([zero1, cum_r])[uo[:-1]]
    q_core = cum_q[last_unit] - q_base
    r_core = cum_r[last_unit] - r_base
    del cum_r, ref_c, qry_c
    a_eff = start + r_core                                                   # reference end of the primary = breakpoint A'
    # ---- the split part: K = 1 (tail) or 2 (mid + tail) segments -------------------------------------------------------------
    two = (kind == 1) | (kind == 7)
    mid_len = torch.where(two, size, torch.zeros_like(size))
    gap = torch.where(kind == 3, size, torch.zeros_like(size))              # read bases between the primary and the tail (split INS)
    tail = rint(400, 3000, R)
    tail = torch.where(kind == 4, tail + size, tail)                        # a tandem copy longer than the duplicated stretch
    trail = torch.where(is_split, mid_len + gap + tail,
                        torch.where(rnd(R) < 0.3, rint(1, 60, R), torch.zeros(R, dtype=i64, device=dev)))
    lseq = lead + q_core + trail
    qa = lead + q_core                                                       # query offset where the primary's alignment ends
    pc, pp = p_contig[pick], p_pos[pick] + rint(-5, 6, R)
    mid_tid = torch.where(kind == 7, pc, r_contig)
    mid_pos = torch.where(kind == 7, pp, a_eff)
    mid_rev = kind == 1
    tail_tid = torch.where(kind == 5, pc, r_contig)
    tail_pos = a_eff.clone()
    tail_pos = torch.where(kind == 1, a_eff + size, tail_pos)
    tail_pos = torch.where(kind == 2, a_eff + size, tail_pos)
    tail_pos = torch.where(kind == 4, (a_eff - size).clamp_min(1), tail_pos)
    tail_pos = torch.where(kind == 5, pp, tail_pos)
    tail_pos = torch.where(kind == 6, a_eff + max_sv_size + 1000 + size, tail_pos)
    tail_q = qa + mid_len + gap                                              # query offset of the tail segment
    # ---- records: primaries + the supplementary records of the split reads, coordinate order -----------------------------------
    sp = torch.nonzero(is_split).flatten()
    sp2 = torch.nonzero(two).flatten()
    NS, N2 = int(sp.numel()), int(sp2.numel())
    n_rec = R + NS + N2
    rec_tid = torch.cat([r_contig, tail_tid[sp], mid_tid[sp2]])
    rec_pos = torch.cat([start, tail_pos[sp], mid_pos[sp2]])
    rec_kind = torch.cat([torch.zeros(R, dtype=i64, device=dev), torch.full((NS,), 2, dtype=i64, device=dev), torch.ones(N2, dtype=i64, device=dev)])   # 0 primary, 1 mid, 2 tail
    rec_read = torch.cat([torch.arange(R, device=dev), sp, sp2])
    rorder = torch.argsort(coff[rec_tid] + rec_pos, stable=True)
    rec_tid, rec_pos, rec_kind, rec_read = rec_tid[rorder], rec_pos[rorder], rec_kind[rorder], rec_read[rorder]
    prim_mask = rec_kind == 0
    rec_of_primary = torch.empty(R, dtype=i64, device=dev)
    rec_of_primary[rec_read[prim_mask]] = torch.nonzero(prim_mask).flatten()
    n_ops_prim = (lead > 0).to(i64) + 2 * n_units + (trail > 0).to(i64)
    rec_nops = torch.where(prim_mask, n_ops_prim[rec_read], torch.where(rec_kind == 1, 3, 2))
    cigar_off = torch.zeros(n_rec + 1, dtype=i64, device=dev)
    torch.cumsum(rec_nops, 0, out=cigar_off[1:])
    n_ops = int(cigar_off[-1].item())
    cigar = torch.zeros(n_ops + 8, dtype=i32, device=dev)
    pbase = cigar_off[rec_of_primary]
    hl = torch.nonzero(lead > 0).flatten()
    cigar[pbase[hl]] = ((lead[hl] << 4) | 4).to(i32)
    ht = torch.nonzero(trail > 0).flatten()
    cigar[pbase[ht] + n_ops_prim[ht] - 1] = ((trail[ht] << 4) | 4).to(i32)
    ur = unit_read.to(i64)
    ubase = (pbase + (lead > 0).to(i64))[ur] + 2 * (torch.arange(U, device=dev) - uo[:-1][ur])
    del ur
    cigar[ubase] = (m_len << 4)
    cigar[ubase + 1] = ((ind_len << 4) | ind_op.to(i32))
    del ubase

    def seg_cigars(dst, off, reads, which):
        """S/M/S CIGARs (reference orientation) of the mid (3 ops) or tail (2 ops) segment of `reads`, written at dst[off...]"""
        if which == 1:                                                       # mid: query [qa, qa + mid_len)
            before, ln = qa[reads], mid_len[reads]
            after = lseq[reads] - before - ln
            rev = mid_rev[reads]
            dst[off] = ((torch.where(rev, after, before) << 4) | 4).to(i32)
            dst[off + 1] = (ln << 4).to(i32)
            dst[off + 2] = ((torch.where(rev, before, after) << 4) | 4).to(i32)
        else:                                                                # tail: query [tail_q, lseq), forward
            dst[off] = ((tail_q[reads] << 4) | 4).to(i32)
            dst[off + 1] = ((lseq[reads] - tail_q[reads]) << 4).to(i32)

    s1 = torch.nonzero(rec_kind == 1).flatten()
    s2 = torch.nonzero(rec_kind == 2).flatten()
    seg_cigars(cigar, cigar_off[s1], rec_read[s1], 1)
    seg_cigars(cigar, cigar_off[s2], rec_read[s2], 2)
    # ---- record metadata --------------------------------------------------------------------------------------------------------
    flag = torch.where(rec_kind == 0, 0, torch.where((rec_kind == 1) & mid_rev[rec_read], 2048 | 16, 2048)).to(torch.int16)
    has_sa = prim_mask & is_split[rec_read]
    flag = torch.where(has_sa, flag | 0x4000, flag)
    mapq = torch.where(rnd(n_rec) < 1.0 - prof["lowq"], 60, 10).to(torch.uint8)
    rec_lseq = lseq[rec_read]
    # ---- sequences (4-bit packed): random bases, planted CIGAR insertions carry their site's sequence + substitutions -----------
    nbytes = (rec_lseq + 1) // 2
    seq_off = torch.zeros(n_rec + 1, dtype=i64, device=dev)
    torch.cumsum(nbytes, 0, out=seq_off[1:])
    total_bytes = int(seq_off[-1].item())
    seq = _rand_acgt_pairs(total_bytes + 16, gen, dev)
    # (read, first inserted read base, length, site) of every planted insertion: CIGAR insertions and the read gap of split-INS reads
    if planted_unit.numel():
        is_ins = s_class[planted_site] == 1
        pu, psite = planted_unit[is_ins], planted_site[is_ins]
        pr = unit_read[pu].to(i64)
        ilen = ind_len[pu].to(i64)
        q0 = lead[pr] + (cum_q[pu] - q_base[pr]) - ilen
    else:
        pr = ilen = q0 = psite = torch.zeros(0, dtype=i64, device=dev)
    r3 = torch.nonzero(kind == 3).flatten()
    pr, ilen, q0, psite = torch.cat([pr, r3]), torch.cat([ilen, size[r3]]), torch.cat([q0, qa[r3]]), torch.cat([psite, pick[r3]])
    if pr.numel():
        byte_lo = q0 // 2
        nb = (q0 + ilen - 1) // 2 - byte_lo + 1
        boff = torch.zeros(pr.numel() + 1, dtype=i64, device=dev)
        torch.cumsum(nb, 0, out=boff[1:])
        TB = int(boff[-1].item())
        run = torch.repeat_interleave(torch.arange(pr.numel(), device=dev), nb)
        k = torch.arange(TB, device=dev) - boff[:-1][run]
        gbyte = seq_off[rec_of_primary[pr]][run] + byte_lo[run] + k
        old = seq[gbyte].to(i64)
        qh = (byte_lo[run] + k) * 2
        out = old
        slot = seq_slot[psite][run]
        for half, shift in ((0, 4), (1, 0)):
            rel = qh + half - q0[run]
            inside = (rel >= 0) & (rel < ilen[run])
            b = site_seq[slot.clamp_min(0), rel.clamp(0, max_ins - 1)].to(i64)
            err = rnd(TB) < prof["ins_err"]
            b = torch.where(err, 1 << rint(0, 4, TB), b)
            nib = torch.where(inside, b, (old >> shift) & 15)
            out = (out & ~(15 << shift)) | (nib << shift)
        seq[gbyte] = out.to(torch.uint8)
    # ---- segment table (SA-derived rows of the split primaries: query order mid, tail) ------------------------------------------
    seg_cnt = torch.where(has_sa, torch.where(two[rec_read], 2, 1), 0)
    seg_off = torch.zeros(n_rec + 1, dtype=i64, device=dev)
    torch.cumsum(seg_cnt, 0, out=seg_off[1:])
    n_seg = int(seg_off[-1].item())
    prim_rec = torch.nonzero(has_sa).flatten()
    pr = rec_read[prim_rec]
    NSG = max(1, n_seg)
    seg_tid = torch.zeros(NSG, dtype=i32, device=dev)
    seg_pos = torch.zeros(NSG, dtype=i32, device=dev)
    seg_rev = torch.zeros(NSG, dtype=torch.uint8, device=dev)
    seg_mapq = torch.full((NSG,), 60, dtype=torch.uint8, device=dev)
    seg_lseq = torch.zeros(NSG, dtype=i32, device=dev)
    seg_cigar_off = torch.zeros(n_seg + 1, dtype=i64, device=dev)
    seg_cigar = torch.zeros(3 * NSG + 8, dtype=i32, device=dev)
    if n_seg:
        sb = seg_off[prim_rec]
        is2 = two[pr]
        tail_row = sb + is2.to(i64)
        cnt = torch.zeros(n_seg, dtype=i64, device=dev)
        cnt[tail_row] = 2
        cnt[sb[is2]] = 3
        torch.cumsum(cnt, 0, out=seg_cigar_off[1:])
        seg_tid[tail_row] = tail_tid[pr].to(i32); seg_pos[tail_row] = tail_pos[pr].to(i32); seg_lseq[tail_row] = lseq[pr].to(i32)
        m2 = pr[is2]
        seg_tid[sb[is2]] = mid_tid[m2].to(i32); seg_pos[sb[is2]] = mid_pos[m2].to(i32); seg_lseq[sb[is2]] = lseq[m2].to(i32)
        seg_rev[sb[is2]] = mid_rev[m2].to(torch.uint8)
        seg_cigars(seg_cigar, seg_cigar_off[sb[is2]], m2, 1)
        seg_cigars(seg_cigar, seg_cigar_off[tail_row], pr, 2)
    genome = _rand_acgt_codes(total, gen, dev)
    # ---- pack -------------------------------------------------------------------------------------------------------------------
    b = DeviceBatch()
    b.n_rec, b.n_seg, b.n_contig = n_rec, n_seg, NC
    b.references = names
    t = b.t
    t["flag"] = flag.view(torch.int16)
    t["tid"] = rec_tid.to(i32)
    t["pos"] = rec_pos.to(i32)
    t["mapq"] = mapq
    t["lseq"] = rec_lseq.to(i32)
    t["read_id"] = rec_read.to(i32)
    t["order"] = (2 * torch.arange(n_rec, device=dev)).to(i32)
    t["seg_order"] = (2 * torch.arange(n_rec, device=dev) + 1).to(i32)
    t["cigar_off"], t["cigar"], t["seq_off"], t["seq"] = cigar_off, cigar, seq_off, seq
    t["seg_off"] = seg_off.to(i32)
    t["seg_tid"], t["seg_pos"], t["seg_rev"], t["seg_mapq"], t["seg_lseq"] = seg_tid, seg_pos, seg_rev, seg_mapq, seg_lseq
    t["seg_cigar_off"], t["seg_cigar"] = seg_cigar_off, seg_cigar
    rank = sorted(range(NC), key=lambda i: names[i])
    cr = [0] * NC
    for r_, i_ in enumerate(rank):
        cr[i_] = r_
    t["contig_rank"] = torch.tensor(cr, dtype=i32, device=dev)
    kinds = torch.bincount(kind, minlength=8).tolist()
    b.meta = dict(workload=prof.get("name", "custom"), n_reads=R, n_records=n_rec, n_ops=n_ops, n_seg=n_seg, n_sites=S, n_contig=NC,
                  contigs=[[n, int(l)] for n, l in prof["contigs"]], genome_bases=total, n_planted=int(planted_unit.numel()),
                  seq_bytes=total_bytes, mean_len=float(L.double().mean().item()), ops_per_read=n_ops / max(1, R),
                  reads_by_layout={KIND_NAMES[i]: int(kinds[i]) for i in range(8)})
    return b, genome, coff, b.meta
