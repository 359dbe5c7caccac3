"""Synthetic ONT-like record batches built with torch ON THE DEVICE (bench.py, large parity tests).

SURVEY.md section 8(d) configuration C2: N reads, log-normal lengths with a given N50, one contig, planted
DEL / INS (inside the CIGAR, +-5 bp size jitter, position jitter from the surrounding match run) and INV sites
(fwd-rev-fwd split reads: the primary ends at the left breakpoint and carries two SA-derived segment rows; the two
supplementary alignments are separate records with 3-op CIGARs).  CIGAR noise follows the reference's own test
generator (src/tests/test_Collect.py:17-46): match runs U[5,30] alternating with 1-3 bp I/D, about one op per 9 bp.

Everything is produced as the Structure-of-Arrays batch of include/svx.h, already resident in HBM (torch tensors);
`Batch.struct()` hands the device pointers to libsvx.  The same code runs on CPU tensors for small tests.
"""
import math

import numpy as np
import torch

from . import _abi



def _rand_acgt_codes(n, gen, dev, chunk=1 << 28):
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        r = torch.randint(0, 4, (hi - lo,), generator=gen, device=dev, dtype=torch.uint8)
        out[lo:hi] = torch.bitwise_left_shift(torch.ones_like(r), r)
    return out


def _rand_acgt_pairs(n, gen, dev, chunk=1 << 28):
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        a = torch.randint(0, 4, (hi - lo,), generator=gen, device=dev, dtype=torch.uint8)
        b = torch.randint(0, 4, (hi - lo,), generator=gen, device=dev, dtype=torch.uint8)
        one = torch.ones_like(a)
        out[lo:hi] = torch.bitwise_left_shift(torch.bitwise_left_shift(one, a), 4) | torch.bitwise_left_shift(one, b)
    return out


class DeviceBatch(object):
    def __init__(self):
        self.t = {}
        self.n_rec = 0
        self.n_seg = 0
        self.n_contig = 1
        self.meta = {}
        self._keep = None

    def struct(self):
        b = _abi.Batch()
        b.on_device = 1 if self.t["cigar"].is_cuda else 0
        b.n_rec, b.n_seg, b.n_contig = self.n_rec, self.n_seg, self.n_contig
        for k in _abi.BATCH_DTYPES:
            setattr(b, k, _abi.ptr(self.t[k]))
        self._keep = b
        return b

    def view_records(self, lo, hi):
        """DeviceBatch of records [lo, hi) that SHARES this batch's arrays: per-record columns and the offset arrays start at record lo, the
        flat arrays (CIGAR, SEQ, segment table) stay whole - offsets are absolute.  Emission slots, read ids and contig ids are unchanged."""
        v = DeviceBatch()
        v.n_rec, v.n_seg, v.n_contig = hi - lo, self.n_seg, self.n_contig
        v.meta = self.meta
        for k, t in self.t.items():
            if k in ("flag", "tid", "pos", "mapq", "lseq", "read_id", "order", "seg_order"):
                v.t[k] = t[lo:hi]
            elif k in ("cigar_off", "seq_off", "seg_off"):
                v.t[k] = t[lo:hi + 1]
            else:
                v.t[k] = t
        if hasattr(self, "references"):
            v.references = self.references
        return v

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.t.values())

    def slice_records(self, lo, hi):
        """Host (numpy) HostBatch-like copy of records [lo, hi) - for the CPU baseline / parity checks."""
        from .batch import HostBatch
        hb = HostBatch()
        t = self.t
        hb.n_rec = hi - lo
        hb.references = list(getattr(self, "references", None) or ["chr1"])
        A = hb.arrays
        for k in ("flag", "tid", "pos", "mapq", "lseq", "read_id", "order", "seg_order"):
            A[k] = t[k][lo:hi].cpu().numpy().copy()
        A["order"] = (A["order"] - A["order"][0] if hi > lo else A["order"]).astype(np.uint32)
        A["seg_order"] = (A["order"] + 1).astype(np.uint32)

        def cut(off_name, data_name, a, b, dtype):
            off = t[off_name][a:b + 1].cpu().numpy().astype(np.int64)
            data = t[data_name][int(off[0]):int(off[-1])].cpu().numpy().copy()
            if data.size == 0:
                data = np.zeros(1, dtype=data.dtype)
            return (off - off[0]).astype(dtype), data
        A["cigar_off"], A["cigar"] = cut("cigar_off", "cigar", lo, hi, np.uint64)
        A["seq_off"], A["seq"] = cut("seq_off", "seq", lo, hi, np.uint64)
        so = t["seg_off"][lo:hi + 1].cpu().numpy().astype(np.int64)
        s0, s1 = int(so[0]), int(so[-1])
        A["seg_off"] = (so - s0).astype(np.uint32)
        hb.n_seg = s1 - s0
        for k in ("seg_tid", "seg_pos", "seg_rev", "seg_mapq", "seg_lseq"):
            v = t[k][s0:s1].cpu().numpy().copy()
            A[k] = v if v.size else np.zeros(1, dtype=v.dtype)
        A["seg_cigar_off"], A["seg_cigar"] = cut("seg_cigar_off", "seg_cigar", s0, s1, np.uint64)
        A["contig_rank"] = t["contig_rank"].cpu().numpy().astype(np.int32).copy()
        hb.read_names = None
        return hb


def _lognormal_lengths(gen, n, n50, device, lo=500, hi=200000):
    # log-normal with sigma 0.8; for a log-normal the N50 (length-weighted median) is exp(mu + sigma^2)
    sigma = 0.8
    mu = math.log(n50) - sigma * sigma
    x = torch.exp(mu + sigma * torch.randn(n, generator=gen, device=device))
    return x.clamp_(lo, hi).to(torch.int64)


def make_batch(n_reads=10000, n50=20000, contig_len=250_000_000, n_sites=None, seed=2, device="cuda",
               frac_del=0.45, frac_ins=0.45, size_lo=50, size_hi=5000, inv_read_frac=0.12, ins_err=0.03,
               max_sites_per_read=6, lengths=None):
    """Returns (DeviceBatch, genome_codes uint8 tensor [contig_len], info dict).
    lengths: None = log-normal with N50 `n50`; ("triangular", lo, hi, mode) = the read-length law of the reference's own test generator
    (src/tests/test_Collect.py:71)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    i64, i32 = torch.int64, torch.int32
    R = int(n_reads)
    if n_sites is None:
        n_sites = max(4, int(25000 * (R / 1e6)))
    # ---- planted sites ---------------------------------------------------------------------------------
    S = int(n_sites)
    site_pos = torch.sort(torch.randint(20000, contig_len - 220000, (S,), generator=gen, device=dev, dtype=i64)).values
    u = torch.rand(S, generator=gen, device=dev)
    site_type = torch.where(u < frac_del, 0, torch.where(u < frac_del + frac_ins, 1, 2)).to(i64)     # 0 DEL 1 INS 2 INV
    site_size = torch.exp(math.log(size_lo) + torch.rand(S, generator=gen, device=dev) * math.log(size_hi / size_lo)).to(i64)
    site_hom = torch.rand(S, generator=gen, device=dev) < 0.5
    max_ins = int(size_hi) + 8
    # per-site inserted sequence (codes 1,2,4,8), only meaningful for INS sites
    site_seq = (1 << torch.randint(0, 4, (S, max_ins), generator=gen, device=dev, dtype=i64)).to(torch.uint8)
    inv_sites = torch.nonzero(site_type == 2).flatten()
    # ---- reads -----------------------------------------------------------------------------------------
    if lengths is not None and lengths[0] == "triangular":
        _, t_lo, t_hi, t_mode = lengths
        uu = torch.rand(R, generator=gen, device=dev, dtype=torch.float64)
        fc = (t_mode - t_lo) / (t_hi - t_lo)
        L = torch.where(uu < fc, t_lo + torch.sqrt(uu * (t_hi - t_lo) * (t_mode - t_lo)),
                        t_hi - torch.sqrt((1 - uu) * (t_hi - t_lo) * (t_hi - t_mode))).to(torch.int64).clamp_min(40)
    else:
        L = _lognormal_lengths(gen, R, n50, dev)                   # aligned query length of the primary (before clips)
    start = torch.randint(1000, contig_len - 210000, (R,), generator=gen, device=dev, dtype=i64)
    is_inv = (torch.rand(R, generator=gen, device=dev) < inv_read_frac) & (inv_sites.numel() > 0)
    inv_pick = inv_sites[torch.randint(0, max(1, inv_sites.numel()), (R,), generator=gen, device=dev)] if inv_sites.numel() else torch.zeros(R, dtype=i64, device=dev)
    inv_a = site_pos[inv_pick] + torch.randint(-5, 6, (R,), generator=gen, device=dev)
    inv_b = site_pos[inv_pick] + site_size[inv_pick] + torch.randint(-5, 6, (R,), generator=gen, device=dev)
    hap = torch.rand(R, generator=gen, device=dev) < 0.5
    lead = torch.where(torch.rand(R, generator=gen, device=dev) < 0.5, torch.randint(1, 60, (R,), generator=gen, device=dev, dtype=i64), 0)
    # ---- CIGAR units: (M run, short indel) --------------------------------------------------------------
    n_units = (L // 19).clamp_min(2)
    uo = torch.zeros(R + 1, dtype=i64, device=dev)
    torch.cumsum(n_units, 0, out=uo[1:])
    U = int(uo[-1].item())
    unit_read = torch.repeat_interleave(torch.arange(R, device=dev, dtype=i32), n_units)
    m_len = torch.randint(5, 31, (U,), generator=gen, device=dev, dtype=i32)
    ind_len = torch.randint(1, 4, (U,), generator=gen, device=dev, dtype=i32)
    ind_op = torch.randint(1, 3, (U,), generator=gen, device=dev, dtype=torch.int8)      # 1 = I, 2 = D
    last_unit = uo[1:] - 1
    ind_op[last_unit] = 0                                                                   # a CIGAR core must not end in I/D
    # an inversion read's primary ends exactly at the (jittered) left breakpoint: fix its start from the reference
    # span of its noise-only CIGAR (such reads receive no planted DEL/INS, so the span is final)
    cum = torch.cumsum(m_len.to(i64) + torch.where(ind_op != 1, ind_len, 0).to(i64), 0)
    r_noise = cum[last_unit] - torch.cat([torch.zeros(1, dtype=i64, device=dev), cum])[uo[:-1]]
    del cum
    start = torch.where(is_inv, (inv_a - r_noise).clamp_min(100), start)
    # ---- plant DEL / INS sites, one per pass, with exact reference coordinates ----------------------------
    planted_unit, planted_site = [], []
    for j in range(max_sites_per_read):
        cum = torch.cumsum(m_len.to(i64) + torch.where(ind_op != 1, ind_len, 0).to(i64), 0)
        base = torch.where(uo[:-1] > 0, cum[(uo[:-1] - 1).clamp_min(0)], torch.zeros_like(uo[:-1]))
        read_ref_end = start + (cum[last_unit] - base)
        lo = torch.searchsorted(site_pos, start + 300)
        hi = torch.searchsorted(site_pos, read_ref_end - 300 - size_hi)
        sidx = lo + j
        ok = sidx < hi
        sidx = sidx.clamp_max(S - 1)
        ok &= (site_type[sidx] != 2) & (site_hom[sidx] | hap) & ~is_inv
        # unit whose match run ends closest below the site position: first unit with (ref pos after its M run) >= target
        target = site_pos[sidx] - start + base                 # in global cumsum coordinates
        after_m = cum - torch.where(ind_op != 1, ind_len, 0).to(i64)      # position after the M run of each unit
        uidx = torch.searchsorted(after_m, target)
        ok &= (uidx > uo[:-1]) & (uidx < last_unit - 1)
        rsel = torch.nonzero(ok).flatten()
        if rsel.numel() == 0:
            continue
        us, ss = uidx[rsel], sidx[rsel]
        jit = torch.randint(-5, 6, (rsel.numel(),), generator=gen, device=dev)
        ind_len[us] = (site_size[ss] + jit).clamp_min(1).to(i32)
        ind_op[us] = torch.where(site_type[ss] == 0, 2, 1).to(torch.int8)
        planted_unit.append(us)
        planted_site.append(ss)
    planted_unit = torch.cat(planted_unit) if planted_unit else torch.zeros(0, dtype=i64, device=dev)
    planted_site = torch.cat(planted_site) if planted_site else torch.zeros(0, dtype=i64, device=dev)
    if planted_unit.numel():
        # a unit planted in two passes keeps the LAST site (its length/op were overwritten in pass order); without this the
        # sequence overwrite below would scatter two different site sequences to the same bytes (order-undefined on a GPU)
        ordk = torch.arange(planted_unit.numel(), device=dev)
        key = planted_unit * (planted_unit.numel() + 1) + ordk
        srt = torch.argsort(key)
        pu_s = planted_unit[srt]
        last = torch.ones_like(pu_s, dtype=torch.bool)
        last[:-1] = pu_s[:-1] != pu_s[1:]
        keep = srt[last]
        planted_unit, planted_site = planted_unit[keep], planted_site[keep]
    # ---- per-read totals ------------------------------------------------------------------------------------
    ref_c = m_len.to(i64) + torch.where(ind_op == 2, ind_len, 0).to(i64) + torch.where(ind_op == 0, ind_len, 0).to(i64)
    qry_c = m_len.to(i64) + torch.where(ind_op == 1, ind_len, 0).to(i64) + torch.where(ind_op == 0, ind_len, 0).to(i64)
    cum_q = torch.cumsum(qry_c, 0)
    cum_r = torch.cumsum(ref_c, 0)
    zero = torch.zeros(1, dtype=i64, device=dev)
    q_base = torch.cat([zero, cum_q])[uo[:-1]]
    r_base = torch.cat([zero, cum_r])[uo[:-1]]
    q_core = cum_q[last_unit] - q_base                       # query bases consumed by the CIGAR core
    r_core = cum_r[last_unit] - r_base
    # inversion reads: trailing soft clip = inverted part + tail
    inv_len = (inv_b - inv_a).clamp_min(40)
    tail = torch.randint(200, 3000, (R,), generator=gen, device=dev, dtype=i64)
    trail = torch.where(is_inv, inv_len + tail, torch.where(torch.rand(R, generator=gen, device=dev) < 0.3,
                                                          torch.randint(1, 60, (R,), generator=gen, device=dev, dtype=i64), 0))
    # the primary of an inversion read ends exactly at its reference end = breakpoint a'
    a_eff = start + r_core
    b_eff = a_eff + inv_len
    lseq = lead + q_core + trail
    # ---- records: primaries + 2 supplementary per inversion read, coordinate order --------------------------
    inv_reads = torch.nonzero(is_inv).flatten()
    NI = int(inv_reads.numel())
    n_rec = R + 2 * NI
    rec_pos = torch.cat([start, a_eff[inv_reads], b_eff[inv_reads]])
    rec_kind = torch.cat([torch.zeros(R, dtype=i64, device=dev), torch.ones(NI, dtype=i64, device=dev), torch.full((NI,), 2, dtype=i64, device=dev)])
    rec_read = torch.cat([torch.arange(R, device=dev), inv_reads, inv_reads])
    rorder = torch.argsort(rec_pos, stable=True)
    rec_pos, rec_kind, rec_read = rec_pos[rorder], rec_kind[rorder], rec_read[rorder]
    rec_of_primary = torch.empty(R, dtype=i64, device=dev)
    prim_mask = rec_kind == 0
    rec_of_primary[rec_read[prim_mask]] = torch.nonzero(prim_mask).flatten()
    n_ops_prim = (lead > 0).to(i64) + 2 * n_units + (trail > 0).to(i64)
    rec_nops = torch.where(prim_mask, n_ops_prim[rec_read], torch.where(rec_kind == 1, 3, 2))
    cigar_off = torch.zeros(n_rec + 1, dtype=i64, device=dev)
    torch.cumsum(rec_nops, 0, out=cigar_off[1:])
    n_ops = int(cigar_off[-1].item())
    cigar = torch.zeros(n_ops + 8, dtype=i32, device=dev)
    pbase = cigar_off[rec_of_primary]                       # first op of each primary
    # lead / trail clips
    hl = torch.nonzero(lead > 0).flatten()
    cigar[pbase[hl]] = ((lead[hl] << 4) | 4).to(i32)
    ht = torch.nonzero(trail > 0).flatten()
    cigar[pbase[ht] + n_ops_prim[ht] - 1] = ((trail[ht] << 4) | 4).to(i32)
    # units
    ur = unit_read.to(i64)
    ubase = (pbase + (lead > 0).to(i64))[ur] + 2 * (torch.arange(U, device=dev) - uo[:-1][ur])
    del ur
    cigar[ubase] = (m_len << 4)                              # op 0 = M
    cigar[ubase + 1] = ((ind_len << 4) | ind_op.to(i32))
    del ubase
    # supplementary records (reverse middle part, forward tail), CIGARs in reference orientation
    s1 = torch.nonzero(rec_kind == 1).flatten()
    s2 = torch.nonzero(rec_kind == 2).flatten()
    r1, r2 = rec_read[s1], rec_read[s2]
    qa = lead + q_core                                        # query offset where the inverted part starts
    # B (reverse strand): leading clip = bases after the segment on the read, trailing clip = bases before it
    o = cigar_off[s1]
    cigar[o] = (((lseq[r1] - qa[r1] - inv_len[r1]) << 4) | 4).to(i32)
    cigar[o + 1] = (inv_len[r1] << 4).to(i32)
    cigar[o + 2] = ((qa[r1] << 4) | 4).to(i32)
    o = cigar_off[s2]
    cigar[o] = (((qa[r2] + inv_len[r2]) << 4) | 4).to(i32)
    cigar[o + 1] = ((lseq[r2] - qa[r2] - inv_len[r2]) << 4).to(i32)
    # ---- record metadata ------------------------------------------------------------------------------------
    flag = torch.where(rec_kind == 0, 0, torch.where(rec_kind == 1, 2048 | 16, 2048)).to(torch.int16)
    # mark primaries that carry SA-derived rows (0x4000)
    has_sa = prim_mask & is_inv[rec_read]
    flag = torch.where(has_sa, flag | 0x4000, flag)
    mapq = torch.where(torch.rand(n_rec, generator=gen, device=dev) < 0.97, 60, 10).to(torch.uint8)
    rec_lseq = lseq[rec_read]
    # ---- sequences (4-bit packed) ----------------------------------------------------------------------------
    nbytes = (rec_lseq + 1) // 2
    seq_off = torch.zeros(n_rec + 1, dtype=i64, device=dev)
    torch.cumsum(nbytes, 0, out=seq_off[1:])
    total_bytes = int(seq_off[-1].item())
    seq = _rand_acgt_pairs(total_bytes + 16, gen, dev)
    # overwrite the planted insertions with their site's sequence (+ substitutions)
    if planted_unit.numel():
        is_ins = site_type[planted_site] == 1
        pu, psite = planted_unit[is_ins], planted_site[is_ins]
        if pu.numel():
            pr = unit_read[pu].to(i64)
            ilen = ind_len[pu].to(i64)
            q0 = lead[pr] + (cum_q[pu] - q_base[pr]) - ilen            # read offset of the first inserted base
            byte_lo = q0 // 2
            byte_hi = (q0 + ilen - 1) // 2
            nb = byte_hi - byte_lo + 1
            boff = torch.zeros(pu.numel() + 1, dtype=i64, device=dev)
            torch.cumsum(nb, 0, out=boff[1:])
            TB = int(boff[-1].item())
            run = torch.repeat_interleave(torch.arange(pu.numel(), device=dev), nb)
            k = torch.arange(TB, device=dev) - boff[:-1][run]
            gbyte = seq_off[rec_of_primary[pr]][run] + byte_lo[run] + k
            old = seq[gbyte].to(i64)
            qh = (byte_lo[run] + k) * 2                                  # read offset of the high nibble
            out = old
            for half, shift in ((0, 4), (1, 0)):
                q = qh + half
                rel = q - q0[run]
                inside = (rel >= 0) & (rel < ilen[run])
                base = site_seq[psite[run], rel.clamp(0, max_ins - 1)].to(i64)
                err = torch.rand(TB, generator=gen, device=dev) < ins_err
                base = torch.where(err, 1 << torch.randint(0, 4, (TB,), generator=gen, device=dev), base)
                nib = torch.where(inside, base, (old >> shift) & 15)
                out = (out & ~(15 << shift)) | (nib << shift)
            seq[gbyte] = out.to(torch.uint8)
    # ---- segment table (SA-derived rows of inversion primaries) ---------------------------------------------
    seg_cnt = torch.where(has_sa, 2, 0)
    seg_off = torch.zeros(n_rec + 1, dtype=i64, device=dev)
    torch.cumsum(seg_cnt, 0, out=seg_off[1:])
    n_seg = int(seg_off[-1].item())
    prim_inv_rec = torch.nonzero(has_sa).flatten()
    pr = rec_read[prim_inv_rec]
    seg_tid = torch.zeros(max(1, n_seg), dtype=i32, device=dev)
    seg_pos = torch.zeros(max(1, n_seg), dtype=i32, device=dev)
    seg_rev = torch.zeros(max(1, n_seg), dtype=torch.uint8, device=dev)
    seg_mapq = torch.full((max(1, n_seg),), 60, dtype=torch.uint8, device=dev)
    seg_lseq = torch.zeros(max(1, n_seg), dtype=i32, device=dev)
    seg_cigar = torch.zeros(max(1, 5 * (n_seg // 2)) + 8, dtype=i32, device=dev)
    seg_cigar_off = torch.zeros(n_seg + 1, dtype=i64, device=dev)
    if n_seg:
        sb = seg_off[prim_inv_rec]
        seg_pos[sb] = a_eff[pr].to(i32); seg_pos[sb + 1] = b_eff[pr].to(i32)
        seg_rev[sb] = 1
        seg_lseq[sb] = lseq[pr].to(i32); seg_lseq[sb + 1] = lseq[pr].to(i32)
        cnt = torch.zeros(n_seg, dtype=i64, device=dev)
        cnt[sb] = 3; cnt[sb + 1] = 2
        torch.cumsum(cnt, 0, out=seg_cigar_off[1:])
        o = seg_cigar_off[sb]
        seg_cigar[o] = (((lseq[pr] - qa[pr] - inv_len[pr]) << 4) | 4).to(i32)
        seg_cigar[o + 1] = (inv_len[pr] << 4).to(i32)
        seg_cigar[o + 2] = ((qa[pr] << 4) | 4).to(i32)
        o = seg_cigar_off[sb + 1]
        seg_cigar[o] = (((qa[pr] + inv_len[pr]) << 4) | 4).to(i32)
        seg_cigar[o + 1] = ((lseq[pr] - qa[pr] - inv_len[pr]) << 4).to(i32)
    # ---- genome ------------------------------------------------------------------------------------------------
    genome = _rand_acgt_codes(contig_len, gen, dev)
    # ---- pack ----------------------------------------------------------------------------------------------------
    b = DeviceBatch()
    b.n_rec, b.n_seg, b.n_contig = n_rec, n_seg, 1
    t = b.t
    t["flag"] = flag.view(torch.int16)
    t["tid"] = torch.zeros(n_rec, dtype=i32, device=dev)
    t["pos"] = rec_pos.to(i32)
    t["mapq"] = mapq
    t["lseq"] = rec_lseq.to(i32)
    t["read_id"] = rec_read.to(i32)
    t["order"] = (2 * torch.arange(n_rec, device=dev)).to(i32)
    t["seg_order"] = (2 * torch.arange(n_rec, device=dev) + 1).to(i32)
    t["cigar_off"] = cigar_off
    t["cigar"] = cigar
    t["seq_off"] = seq_off
    t["seq"] = seq
    t["seg_off"] = seg_off.to(i32)
    t["seg_tid"], t["seg_pos"], t["seg_rev"], t["seg_mapq"], t["seg_lseq"] = seg_tid, seg_pos, seg_rev, seg_mapq, seg_lseq
    t["seg_cigar_off"] = seg_cigar_off
    t["seg_cigar"] = seg_cigar
    t["contig_rank"] = torch.zeros(1, dtype=i32, device=dev)
    b.meta = dict(n_reads=R, n_records=n_rec, n_ops=n_ops, n_seg=n_seg, n_sites=S, n_inv_reads=NI, contig_len=contig_len,
                  n_planted=int(planted_unit.numel()), seq_bytes=total_bytes,
                  n50_target=n50, mean_len=float(L.double().mean().item()))
    return b, genome, b.meta
