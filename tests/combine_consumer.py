"""Test infrastructure: a compact restatement of how SVIM's COMBINE step CONSUMES the six cluster lists CLUSTER returns
(src/svim/SVIM_merging.py:12-29, :93-159 and src/svim/SVIM_COMBINE.py:332-478 with --skip_consensus).

COMBINE itself is out of scope (host Python, kept as it is); what matters for a drop-in is that its access
pattern works on svim_amd's lazy lists and objects: attribute reads (`members`, `score`, `std_span`, `std_pos`, `direction1/2`,
`get_source()`, `get_destination()`, `get_key()`), `+` on member lists, `extend()` on the breakend list and on the
insertion-from list, `del lst[i]` on the insertion list.  The reference is not present on the GPU box, so this module replays
those operations - the expected rows in tests/golden/g_combine.json.gz come from the reference's own functions
(tests/golden/make_golden.py:gen_combine, which also ran the reference's code on our lists at generation time).
Rows, not objects, are compared: every function here returns plain dicts / lists.
"""
from bisect import bisect_left

from svim_amd.signatures import SignatureClusterBiLocal
from svim_amd.SVIM_clustering import partition_and_cluster_candidates, span_position_distance_clusters


class Cand(object):
    """stand-in for svim.SVCandidate.CandidateDuplicationInterspersed (same constructor order, src/svim/SVCandidate.py:425-448)"""
    type = "DUP_INT"

    def __init__(self, source_contig, source_start, source_end, dest_contig, dest_start, dest_end, members, score, std_span, std_pos,
                 cutpaste=False):
        self.source_contig, self.source_start, self.source_end = source_contig, max(0, source_start), source_end
        self.dest_contig, self.dest_start, self.dest_end = dest_contig, max(0, dest_start), dest_end
        self.members, self.score, self.std_span, self.std_pos, self.cutpaste = members, score, std_span, std_pos, cutpaste

    def get_source(self):
        return (self.source_contig, self.source_start, self.source_end)

    def get_destination(self):
        return (self.dest_contig, self.dest_start, self.dest_end)

    def get_key(self):
        return (self.type, self.source_contig, self.source_end)

    def downstream_distance_to(self, other):
        if self.type == other.type and self.source_contig == other.source_contig:
            return max(0, other.source_start - self.source_end)
        return float("inf")


def _row(kind, idx, members, **kw):
    d = dict(kw)
    d["members"] = [idx[id(m)] for m in members]
    d["class"] = kind
    d.setdefault("support_fraction", ".")
    d.setdefault("genotype", "./.")
    d.setdefault("ref_reads", None)
    d.setdefault("alt_reads", None)
    return d


def _dup_int_row(c, idx):
    return _row("CandidateDuplicationInterspersed", idx, c.members, source_contig=c.source_contig, source_start=c.source_start,
                source_end=c.source_end, dest_contig=c.dest_contig, dest_start=c.dest_start, dest_end=c.dest_end, score=c.score,
                std_span=c.std_span, std_pos=c.std_pos, cutpaste=c.cutpaste, type="DUP_INT")


def closest_index(sorted_values, x):
    """index of the value nearest to x, the lower one on a tie (src/svim/SVIM_merging.py:32-50)"""
    k = bisect_left(sorted_values, x)
    if k == 0:
        return 0
    if k == len(sorted_values):
        return k - 1
    return k if sorted_values[k] - x < x - sorted_values[k - 1] else k - 1


def merged_score(main, dists, stds, dest_stds):
    """src/svim/SVIM_merging.py:57-90"""
    f = 1.0
    for d in dists:
        f *= max(0, 100 - d) / 100
    for s in list(stds) + list(dest_stds):
        f *= 1 if s is None else max(0, 100 - s) / 100
    return pow(f, 1 / 6) * main


def merge_breakends_at_insertions(bnd_clusters, ins_clusters, options):
    """src/svim/SVIM_merging.py:93-159: EXTENDS bnd_clusters by the mirrored clusters, returns (new DUP_INT clusters, insertion indices)"""
    if len(ins_clusters) == 0:
        return [], []
    mirrored = []
    for c in bnd_clusters:
        m = SignatureClusterBiLocal(c.dest_contig, c.dest_start, c.dest_end, c.source_contig, c.source_start, c.source_end, c.score, c.size,
                                    c.members, c.type, c.std_pos, c.std_span)
        m.direction1 = "fwd" if c.direction2 == "rev" else "rev"
        m.direction2 = "fwd" if c.direction1 == "rev" else "rev"
        mirrored.append(m)
    bnd_clusters.extend(mirrored)
    by = {"fwd": {}, "rev": {}}
    for c in bnd_clusters:
        if c.direction1 == c.direction2:
            by[c.direction1].setdefault(c.source_contig, []).append(c)
    starts = {}
    for d in by:
        for contig in by[d]:
            by[d][contig] = sorted(by[d][contig], key=lambda c: c.get_key())
        starts[d] = {contig: [c.source_start for c in lst] for contig, lst in by[d].items()}
    new_clusters, remove = [], []
    for k, ins in enumerate(ins_clusters):
        contig, s, e = ins.get_source()
        if contig not in starts["fwd"]:
            continue
        kf = closest_index(starts["fwd"][contig], s)
        if contig not in starts["rev"]:
            continue
        kr = closest_index(starts["rev"][contig], s)
        df, dr = abs(starts["fwd"][contig][kf] - s), abs(starts["rev"][contig][kr] - s)
        if df > options.trans_sv_max_distance or dr > options.trans_sv_max_distance:
            continue
        cf, cr = by["fwd"][contig][kf], by["rev"][contig][kr]
        dist = abs(cr.dest_start - cf.dest_start)
        if cr.dest_contig != cf.dest_contig or not 0.95 <= (e - s + 1) / (dist + 1) <= 1.1:
            continue
        members = ins.members + cf.members + cr.members
        score = merged_score(ins.score, [df, dr], [cf.std_span, cr.std_span], [cf.std_pos, cr.std_pos])
        new_clusters.append(SignatureClusterBiLocal(cr.dest_contig, min(cr.dest_start, cf.dest_start), max(cr.dest_start, cf.dest_start), contig, s,
                                                    s + dist, score, len(members), members, "DUP_INT", ins.std_span, ins.std_pos))
        remove.append(k)
    return new_clusters, remove


def flag_cutpaste(dup_int_clusters, del_clusters, options):
    """src/svim/SVIM_merging.py:12-29"""
    out = []
    for c in dup_int_clusters:
        best = min(span_position_distance_clusters(d, c, options.position_distance_normalizer) for d in del_clusters)
        (sc, ss, se), (dc, ds, de) = c.get_source(), c.get_destination()
        out.append(Cand(sc, ss, se, dc, ds, de, c.members, c.score, c.std_span, c.std_pos, cutpaste=best <= options.del_ins_dup_max_distance))
    return out


def consume(clusters6, options, idx):
    """Everything the golden's `expected` block holds, from the 6-tuple of cluster lists (idx: id(signature object) -> row index)."""
    import copy
    dele, insr, inv, tan, dint, bnd = clusters6
    b2, i2 = copy.copy(bnd), copy.copy(insr)
    new_from, to_remove = merge_breakends_at_insertions(b2, i2, options)
    out = {"merged_insertion_from_clusters": [[c.source_contig, c.source_start, c.source_end, c.dest_contig, c.dest_start, c.dest_end, c.score, c.size,
                                               [idx[id(m)] for m in c.members], c.type, c.std_span, c.std_pos] for c in new_from],
           "inserted_regions_to_remove": to_remove, "n_bnd_after_merge": len(b2),
           "flag_cutpaste": [_dup_int_row(c, idx) for c in flag_cutpaste(list(dint) + new_from, dele, options)], "n_ins_before": len(insr)}
    # ---- combine_clusters (src/svim/SVIM_COMBINE.py:332-478) ----
    inv_rows = [_row("CandidateInversion", idx, c.members, source_contig=c.contig, source_start=max(0, c.start), source_end=c.end, score=c.score,
                     std_span=c.std_span, std_pos=c.std_pos, type="INV") for c in inv]
    tan_cands = []
    for c in tan:
        (sc, ss, se), (_, ds, de) = c.get_source(), c.get_destination()
        tan_cands.append((c, int(round((de - ds) / (se - ss))), bool(sum(m.fully_covered for m in c.members))))
    bnd_rows = [_row("CandidateBreakend", idx, c.members, source_contig=c.source_contig, source_start=max(0, c.source_start), source_direction=c.direction1,
                     dest_contig=c.dest_contig, dest_start=max(0, c.dest_start), dest_direction=c.direction2, score=c.score, std_pos1=c.std_span,
                     std_pos2=c.std_pos, type="BND") for c in bnd]
    new_from, remove1 = merge_breakends_at_insertions(bnd, insr, options)
    dint.extend(new_from)
    dup_cands = flag_cutpaste(dint, dele, options)
    # insertions that coincide with an interspersed or a tandem duplication (two merged walks over sorted destinations, :404-452)
    it_int = iter(sorted(dup_cands, key=lambda c: c.get_destination()))
    it_tan = iter(sorted(((c.source_contig, c.source_end, c.source_end + n * (c.source_end - max(0, c.source_start))) for c, n, _ in tan_cands)))
    cur_int = next(it_int, None)
    cur_tan = next(it_tan, None)
    remove2 = []
    for k, ins in enumerate(insr):
        c1, s1, e1 = ins.get_source()
        len1 = e1 - s1
        if cur_int is not None:
            c2, s2, e2 = cur_int.get_destination()
            while c2 < c1 or (c2 == c1 and e2 < s1):
                cur_int = next(it_int, None)
                if cur_int is None:
                    break
                c2, s2, e2 = cur_int.get_destination()
        if cur_int is not None:
            len2 = e2 - s2
            if c2 == c1 and s2 < e1 and (len1 - len2) / max(len1, len2) < 0.2:
                remove2.append(k)
        else:
            if cur_tan is not None:
                c2, s2, e2 = cur_tan
                while c2 < c1 or (c2 == c1 and e2 < s1):
                    cur_tan = next(it_tan, None)
                    if cur_tan is None:
                        break
                    c2, s2, e2 = cur_tan
            if cur_tan is not None:
                len2 = e2 - s2
                if c2 == c1 and s2 < e1 and (len1 - len2) / max(len1, len2) < 0.2:
                    remove2.append(k)
    for k in sorted(set(remove1 + remove2), reverse=True):
        del insr[k]
    del_rows = [_row("CandidateDeletion", idx, c.members, source_contig=c.contig, source_start=max(0, c.start), source_end=c.end, score=c.score,
                     std_span=c.std_span, std_pos=c.std_pos, type="DEL") for c in dele if c.score > 0]
    ins_rows = [_row("CandidateNovelInsertion", idx, c.members, dest_contig=c.contig, dest_start=max(0, c.start), dest_end=c.end, sequence="", score=c.score,
                     std_span=c.std_span, std_pos=c.std_pos, type="INS") for c in insr if c.score > 0]
    final_dups = partition_and_cluster_candidates(dup_cands, options, "interspersed duplication candidates")
    tan_rows = [_row("CandidateDuplicationTandem", idx, c.members, source_contig=c.source_contig, source_start=max(0, c.source_start), source_end=c.source_end,
                     copies=n, score=c.score, std_span=c.std_span, std_pos=c.std_pos, type="DUP_TAN", fully_covered=fc) for c, n, fc in tan_cands]
    out.update({"n_ins_after": len(insr), "n_dup_int_after": len(dint),
                "combine": [del_rows, inv_rows, [_dup_int_row(c, idx) for c in final_dups], tan_rows, ins_rows, bnd_rows]})
    return out
