"""Contig-sharded multi-GPU COLLECT+CLUSTER: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  SURVEY.md section 8(e), DESIGN.md section 6.

Ownership (round 6: coordinate WINDOWS).  The partition keys of a type are sorted by (contig name, key coordinate) - `end`, for INS `start`, for BND `pos1`
(src/svim/SVSignature.py get_key) - and ranks own CONSECUTIVE ranges of that order: world - 1 cuts (contig, coordinate), class Windows.  A cut inside a
contig is only legal where no partition can straddle it: form_partitions (src/svim/SVIM_clustering.py:17-29) cuts where the gap to the previous element
exceeds partition_max_distance, so a cut goes into a CORRIDOR - a stretch wider than partition_max_distance that no signature's [start, end] of any type
touches.  assign_windows proposes cuts that balance a weight (contig length, or a density histogram); Windows.refine moves every proposal into the nearest
corridor from the signatures the ranks actually collected (one small all-gather of merged intervals per step) or, where there is none, to the contig's
edge.  Whole-contig ownership (assign_contigs, rounds 2-5) is the special case of cuts at contig starts.  DUP_INT rows - keyed (destination contig,
SOURCE contig, destination start), i.e. not in coordinate order inside a contig - stay whole-contig: they belong to the rank that owns the destination
contig's first base.  Every partition is local to one rank, and the global sorted order of all partitions of a type is rank-major.

What crosses the fabric per step
  1. foreign signatures: a read's record is collected by the rank that owns the record's contig, but a split read can emit a
     signature that belongs to another contig's owner (a BND whose canonical first end lies elsewhere, a DUP_INT inserted
     elsewhere, a DEL between two supplementary segments on another contig).  Those rows - typically none to a few per
     thousand - are exchanged (one count exchange; nothing else when every count is zero).
  2. the random.sample stream positions.  The stream of a type is consumed by its > 100-member partitions in global order without
     re-seeding (SVIM_clustering.py:129-134): rank r continues where ranks 0..r-1 stop.  svx_cluster finds its start positions itself
     with all-gathers only (svx_cluster_set_ranks; TorchAllGather is the transport): the sizes of everybody's large partitions, then every
     rank's transfer table "position before my partitions -> position after them" - no rank waits for another rank's sampling.
  3. the final candidate gather to rank 0: cluster records, member lists, and the fixed-width signature columns.
Signature columns of ordinary (non-foreign) signatures and inserted sequences never leave their rank before the final gather.

The same code runs on CPU tensors over gloo with the oracle as stand-in engine (tests/test_multigpu_gloo.py).
"""
import os
import time

import numpy as np

from . import _abi
from ._abi import CLU_DTYPES, SIG_DTYPES

SIG_COLS = ("key", "type", "src", "aux", "contig", "start", "end", "contig2", "pos2", "read_id")


def assign_contigs(names, lengths, world):
    """owner rank of every contig: the name-sorted contig list is cut into `world` consecutive ranges of about equal total length
    (a contig goes to the range its midpoint falls into).  Returns int32 [n_contig]."""
    n = len(names)
    order = sorted(range(n), key=lambda i: names[i])
    total = float(sum(lengths)) or 1.0
    owner = np.zeros(n, dtype=np.int32)
    acc = 0.0
    for i in order:
        mid = acc + lengths[i] / 2.0
        owner[i] = min(world - 1, int(mid * world / total))
        acc += lengths[i]
    # ranges are monotone in name order (midpoints are); a rank whose range holds no contig midpoint owns nothing and only takes part in the exchanges
    return owner


def owner_contig(typ, contig, contig2):
    """contig whose owner clusters the signature (torch tensors or numpy arrays)"""
    is_dup_int = typ == _abi.SVX_DUP_INT
    if isinstance(typ, np.ndarray):
        return np.where(is_dup_int, contig2, contig)
    import torch
    return torch.where(is_dup_int, contig2, contig)


def _is_np(x):
    return isinstance(x, np.ndarray)


class Windows(object):
    """Coordinate-window ownership: rank r owns the keys (contig name rank, coordinate) in [cut[r-1], cut[r]) (cut[-1] = -inf, cut[world-1] = +inf).

    crank        int32 [n_global]: rank of every contig NAME in str order (batch.contig_ranks)
    cut_contig   int   [world - 1]: global contig id of every cut
    cut_pos      int64 [world - 1]: coordinate of the cut on that contig; -1 = the contig's first base belongs to the rank above the cut (whole contigs)
    A key is the pair packed into one int64: crank << 33 | (coordinate + 1) - coordinates are < 2^31."""

    def __init__(self, world, crank, cut_contig, cut_pos, refined=False):
        self.world = int(world)
        self.crank = np.asarray(crank, dtype=np.int64)
        self.cut_contig = np.asarray(cut_contig, dtype=np.int64).reshape(-1)
        self.cut_pos = np.asarray(cut_pos, dtype=np.int64).reshape(-1)
        assert self.cut_contig.size == self.world - 1 and self.cut_pos.size == self.world - 1
        self.refined = refined or bool((self.cut_pos < 0).all())
        keys = self._cut_keys()
        assert (np.diff(keys) >= 0).all(), "cuts must be monotone in (contig name, coordinate) order"

    def _cut_keys(self):
        return (self.crank[self.cut_contig] << 33) | (self.cut_pos + 1).clip(0)

    @classmethod
    def from_contig_owner(cls, owner, crank):
        """whole-contig ownership (assign_contigs) as cuts: the cut below rank r sits at the start of the first contig (name order) of a rank >= r"""
        owner = np.asarray(owner)
        crank = np.asarray(crank, dtype=np.int64)
        world = int(owner.max()) + 1 if owner.size else 1
        return cls._whole(owner, crank, world)

    @classmethod
    def _whole(cls, owner, crank, world):
        order = np.argsort(crank, kind="stable")                    # contig ids in name order
        cut_contig, cut_pos = [], []
        for r in range(1, world):
            later = [int(i) for i in order if owner[i] >= r]
            if later:
                cut_contig.append(later[0]); cut_pos.append(-1)
            else:                                                    # nothing at or above r: the cut sits behind everything
                cut_contig.append(int(order[-1])); cut_pos.append((1 << 32) - 2)
        return cls(world, crank, cut_contig, cut_pos, refined=True)

    def with_world(self, world):
        return self

    # ---- who owns a key ---------------------------------------------------------------------------------------------------
    def _rank_of_keys(self, key):
        cuts = self._cut_keys()
        if _is_np(key):
            return np.searchsorted(cuts, key, side="right").astype(np.int64)
        import torch
        return torch.searchsorted(torch.as_tensor(cuts, device=key.device), key, right=True)

    def owner_of_positions(self, contig_gid, pos):
        """owner of (global contig id, coordinate) pairs: numpy arrays or torch tensors"""
        if _is_np(contig_gid):
            key = (self.crank[np.asarray(contig_gid, dtype=np.int64)] << 33) | (np.asarray(pos, dtype=np.int64) + 1).clip(0)
            return self._rank_of_keys(key)
        import torch
        cr = torch.as_tensor(self.crank, device=contig_gid.device)
        key = (cr[contig_gid.long()] << 33) | (pos.long() + 1).clamp_min(0)
        return self._rank_of_keys(key)

    def owner_of_signatures(self, typ, contig_gid, contig2_gid, start, end, pos2):
        """owner of signature rows (columns of a signature table with GLOBAL contig ids): the coordinate is the one the partition key uses
        (csrc/cluster.hip k_make_keys); DUP_INT rows go with the first base of their destination contig"""
        if _is_np(typ):
            t = typ.astype(np.int64)
            is_di = t == _abi.SVX_DUP_INT
            c = np.where(is_di, contig2_gid, contig_gid).astype(np.int64)
            coord = np.where(t == _abi.SVX_INS, start, np.where(t == _abi.SVX_BND, start, end)).astype(np.int64)
            coord = np.where(is_di, -1, coord)
            return self.owner_of_positions(c.clip(0), coord)
        import torch
        t = typ.long()
        is_di = t == _abi.SVX_DUP_INT
        c = torch.where(is_di, contig2_gid.long(), contig_gid.long()).clamp_min(0)
        coord = torch.where((t == _abi.SVX_INS) | (t == _abi.SVX_BND), start.long(), end.long())
        coord = torch.where(is_di, torch.full_like(coord, -1), coord)
        return self.owner_of_positions(c, coord)

    # ---- moving proposed cuts into corridors ------------------------------------------------------------------------------
    def needs_refine(self):
        return not self.refined

    def local_intervals(self, typ, contig_gid, start, end, radius):
        """per cut that lies inside a contig: this rank's signature intervals [start, end] (DUP_INT excluded) within `radius` of the proposal, MERGED -
        a flat int64 array [cut index, lo, hi]* (numpy).  What the ranks all-gather before refine_from."""
        out = []
        typ = np.asarray(typ); contig_gid = np.asarray(contig_gid); start = np.asarray(start, dtype=np.int64); end = np.asarray(end, dtype=np.int64)
        for k in range(self.world - 1):
            x = int(self.cut_pos[k])
            if x < 0:
                continue
            sel = (contig_gid == self.cut_contig[k]) & (typ != _abi.SVX_DUP_INT) & (np.maximum(start, end) >= x - radius) & (np.minimum(start, end) <= x + radius)
            if not sel.any():
                continue
            lo, hi = np.minimum(start[sel], end[sel]), np.maximum(start[sel], end[sel])
            o = np.argsort(lo, kind="stable")
            lo, hi = lo[o], hi[o]
            run_hi = np.maximum.accumulate(hi)
            new = np.concatenate([[True], lo[1:] > run_hi[:-1]])                      # an interval that starts beyond everything before it opens a merged one
            first = np.nonzero(new)[0]
            last = np.concatenate([first[1:] - 1, [lo.size - 1]])
            for a, b in zip(lo[first], run_hi[last]):
                out += [k, int(a), int(b)]
        return np.asarray(out, dtype=np.int64)

    def refine_from(self, gathered, max_distance, radius, lengths=None):
        """gathered: concatenation of every rank's local_intervals.  Every proposal inside a contig moves to the START of the nearest corridor's right edge -
        the first covered coordinate behind a gap wider than max_distance that lies within `radius` - or, where the neighbourhood holds no such gap, down to the
        contig's first base (the contig then belongs whole to the rank above the cut; monotonicity is restored by pushing later cuts of the same contig along).
        Deterministic: all ranks compute the same cuts from the same gathered bytes."""
        g = np.asarray(gathered, dtype=np.int64).reshape(-1, 3)
        new_pos = self.cut_pos.copy()
        for k in range(self.world - 1):
            x = int(self.cut_pos[k])
            if x < 0:
                continue
            iv = g[g[:, 0] == k][:, 1:]
            if iv.shape[0] == 0:
                continue                                                              # nobody has a signature near the proposal: it is in a corridor already
            o = np.argsort(iv[:, 0], kind="stable")
            lo, hi = iv[o, 0], np.maximum.accumulate(iv[o, 1])
            # gaps between consecutive merged stretches (and the open ends of the neighbourhood, as far as it was looked at)
            edges_l = np.concatenate([[x - radius - max_distance - 2], hi])             # left edge of every gap = right end of what lies before it
            edges_r = np.concatenate([lo, [x + radius + max_distance + 2]])             # right edge = first covered coordinate behind it
            ok = (edges_r - edges_l) > max_distance
            # a gap is usable if its left neighbour really ends before its right neighbour starts (overlapping stretches produce negative gaps)
            if not ok.any():
                new_pos[k] = -1
                continue
            cand = np.nonzero(ok)[0]
            # the cut goes to the right edge R of the gap (left rows end <= L < R <= right rows' start); distance of the proposal to the gap
            dist = np.where(edges_r[cand] < x, x - edges_r[cand], np.where(edges_l[cand] > x, edges_l[cand] - x, 0))
            j = cand[int(np.argmin(dist))]
            r_edge = int(edges_r[j])
            if j == edges_r.size - 1:                                                 # the open gap behind the last stretch: anywhere behind it, keep the proposal if it is inside
                r_edge = max(x, int(edges_l[j]) + 1)
            elif j == 0 and x <= int(edges_r[0]):
                r_edge = min(max(x, 0), int(edges_r[0]))                              # the open gap in front of the first stretch
            new_pos[k] = max(r_edge, 0)
        # monotone again inside every contig (a cut that fell back to -1 takes the cuts before it on the same contig with it)
        for k in range(self.world - 2, -1, -1):
            if k + 1 < self.world - 1 and self.cut_contig[k] == self.cut_contig[k + 1] and new_pos[k] > new_pos[k + 1]:
                new_pos[k] = new_pos[k + 1]
        return Windows(self.world, self.crank, self.cut_contig, new_pos, refined=True)


def assign_windows(names, lengths, world, weights=None, bin_size=1 << 20):
    """Proposed cuts that give every rank about the same WEIGHT of the (name-sorted contig, coordinate) order: weights None -> bases (contig lengths); else
    weights[i] = array of per-bin weights of contig i (bins of bin_size bases: records or signatures per bin from a first look at the input).  Returns Windows
    (not refined: cluster_step moves the cuts into corridors)."""
    n = len(names)
    order = sorted(range(n), key=lambda i: names[i])
    crank = np.zeros(n, dtype=np.int64)
    crank[order] = np.arange(n)
    if weights is None:
        weights = [np.full(max(1, -(-int(lengths[i]) // bin_size)), float(bin_size)) for i in range(n)]
        for i in range(n):
            if lengths[i] % bin_size:
                weights[i][-1] = float(lengths[i] % bin_size)
    cum = []                       # (contig, bin, cumulative weight before the bin, weight)
    total = 0.0
    for i in order:
        w = np.asarray(weights[i], dtype=np.float64)
        cum.append((i, total + np.concatenate([[0.0], np.cumsum(w)[:-1]]), w))
        total += float(w.sum())
    cut_contig, cut_pos = [], []
    for r in range(1, world):
        target = total * r / world
        done = False
        for i, before, w in cum:
            end = before[-1] + w[-1] if w.size else before[0] if before.size else 0.0
            if w.size and target < end:
                b = int(np.searchsorted(before, target, side="right") - 1)
                frac = (target - before[b]) / w[b] if w[b] > 0 else 0.0
                bin_len = min(bin_size, int(lengths[i]) - b * bin_size)                  # (the last bin of a contig is shorter)
                pos = b * bin_size + int(frac * bin_len)
                pos = min(pos, int(lengths[i]) - 1)
                cut_contig.append(i); cut_pos.append(pos if pos > 0 else -1)
                done = True
                break
        if not done:
            cut_contig.append(order[-1]); cut_pos.append((1 << 32) - 2)
    return Windows(world, crank, cut_contig, cut_pos, refined=False)


class TorchAllGather(object):
    """The transport svx_cluster_set_ranks asks for: all-gather of a small byte string over torch.distributed (one all_gather_into_tensor;
    "nccl" = RCCL on device tensors, gloo on CPU tensors)."""

    def __init__(self, world, device="cpu"):
        self.world, self.device = world, device
        self.calls, self.bytes = 0, 0

    def __call__(self, send):
        import torch
        import torch.distributed as dist
        n = len(send)
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        t = torch.frombuffer(bytearray(send), dtype=torch.uint8).to(dev)
        out = torch.empty(n * self.world, dtype=torch.uint8, device=dev)
        with _Timed("all_gather(rank exchange of svx_cluster)", n * self.world):
            dist.all_gather_into_tensor(out, t)
        self.calls += 1
        self.bytes += n * self.world
        return out.cpu().numpy().tobytes()

    def gather_into(self, send_addr, recv_addr, nbytes):
        """the same on libsvx's own host buffers (Engine.set_ranks prefers this form): the collective reads `nbytes` at send_addr and writes
        nbytes * world at recv_addr - over gloo with no copy at all, over RCCL through one staging tensor each way (the buffers are host memory)"""
        import ctypes
        import torch
        import torch.distributed as dist
        src = torch.frombuffer((ctypes.c_uint8 * nbytes).from_address(send_addr), dtype=torch.uint8)
        dst = torch.frombuffer((ctypes.c_uint8 * (nbytes * self.world)).from_address(recv_addr), dtype=torch.uint8)
        with _Timed("all_gather(rank exchange of svx_cluster)", nbytes * self.world):
            if dist.get_backend() == "nccl":
                out = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
                dist.all_gather_into_tensor(out, src.to(self.device))
                dst.copy_(out)
            else:
                dist.all_gather_into_tensor(dst, src)
        self.calls += 1
        self.bytes += nbytes * self.world


class _CountingRandom(__import__("random").Random):
    """CPython's own generator (the one the reference uses) counting the 32-bit words it hands out: every getrandbits(k <= 32) is one word"""
    words = 0

    def getrandbits(self, k):
        self.words += 1
        return super().getrandbits(k)


def stream_words_after(sizes):
    """32-bit words random.seed(1524) ... random.sample(range(n), 100) for n in sizes ... has consumed (src/svim/SVIM_clustering.py:129-134)"""
    r = _CountingRandom(1524)
    for n in sizes:
        r.sample(range(int(n)), 100)
    return r.words


# What crosses the fabric: every collective of this module is counted here (payload bytes as the wire sees them: all-gathers count world x the padded slot,
# gathers to rank 0 the padded slots of all ranks).  SVX_WIRE_STATS=1 also times them (a device synchronise on both sides of every collective: the timed region
# of a bench run is perturbed by it, so it is off by default).  bench.py prints the totals of the last step; DESIGN.md section 6 has the expected sizes.
WIRE = {"collectives": 0, "bytes": 0, "seconds": 0.0, "by_kind": {}}


def wire_reset():
    WIRE.update(collectives=0, bytes=0, seconds=0.0, by_kind={})


def _wire_count(kind, nbytes, seconds=0.0):
    WIRE["collectives"] += 1
    WIRE["bytes"] += int(nbytes)
    WIRE["seconds"] += seconds
    k = WIRE["by_kind"].setdefault(kind, [0, 0, 0.0])
    k[0] += 1
    k[1] += int(nbytes)
    k[2] += seconds


class _Timed(object):
    """with _Timed(kind, bytes): <collective> - counts it; with SVX_WIRE_STATS=1 also its wall time between two device synchronises"""

    def __init__(self, kind, nbytes):
        self.kind, self.nbytes = kind, nbytes
        self.on = os.environ.get("SVX_WIRE_STATS") == "1"

    def __enter__(self):
        if self.on:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        dt = 0.0
        if self.on:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = time.perf_counter() - self.t0
        _wire_count(self.kind, self.nbytes, dt)
        return False


def barrier():
    """dist.barrier that names this rank's device under RCCL (otherwise torch guesses it from the rank)"""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def _wire(device):
    """device the collectives run on: the tensors' own under RCCL; the host under gloo (a two-process test on one GPU stages through it)"""
    import torch.distributed as dist
    return device if dist.get_backend() == "nccl" else "cpu"


def _all_gather_counts(values, device):
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    w = _wire(device)
    t = torch.tensor(values, dtype=torch.int64, device=w)
    out = torch.zeros(world * len(values), dtype=torch.int64, device=w)
    with _Timed("all_gather(counts)", out.numel() * 8):
        dist.all_gather_into_tensor(out, t)
    return out.view(world, len(values)).tolist()


def _all_gather_rows(t, counts):
    """all-gather of 1-D tensors with per-rank lengths `counts` (known from ONE count exchange) -> list of per-rank tensors"""
    import torch
    import torch.distributed as dist
    mx = max(max(counts), 1)
    w = _wire(t.device)
    pad = torch.zeros(mx, dtype=t.dtype, device=w)
    pad[:t.numel()] = t
    out = torch.empty(mx * len(counts), dtype=t.dtype, device=w)
    with _Timed("all_gather(foreign rows)", out.numel() * out.element_size()):
        dist.all_gather_into_tensor(out, pad)
    out = out.to(t.device)
    return [out[r * mx:r * mx + c] for r, c in enumerate(counts)]


def _all_gather_table(cols, names, counts):
    """One all-gather for a whole table: the columns `names` of `cols` (1-D tensors of this rank's row count, any dtypes) travel as ONE byte buffer (column after
    column, padded to the largest rank) instead of one collective per column -> {name: [per-rank tensors]}"""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    n = counts[rank]
    sizes = [cols[k].element_size() for k in names]
    dtypes = [cols[k].dtype for k in names]
    row = sum(sizes)
    dev = cols[names[0]].device
    mine = torch.cat([cols[k].contiguous().view(torch.uint8) for k in names]) if n else torch.zeros(0, dtype=torch.uint8, device=dev)
    got = _all_gather_rows(mine, [c * row for c in counts])
    out = {k: [] for k in names}
    for r, c in enumerate(counts):
        off = 0
        for k, sz, dt in zip(names, sizes, dtypes):
            out[k].append(got[r][off:off + c * sz].clone().view(dt))          # clone: a fresh, aligned buffer for the typed view
            off += c * sz
    return out


def _gather_to_root(t, counts, rank):
    """dist.gather of 1-D tensors with per-rank lengths `counts` to rank 0 -> concatenation (rank 0) / None"""
    import torch
    import torch.distributed as dist
    mx = max(max(counts), 1)
    w = _wire(t.device)
    pad = torch.zeros(mx, dtype=t.dtype, device=w)
    pad[:t.numel()] = t
    with _Timed("gather(final, to rank 0)", mx * len(counts) * pad.element_size()):
        if rank == 0:
            bufs = [torch.empty(mx, dtype=t.dtype, device=w) for _ in counts]
            dist.gather(pad, bufs, dst=0)
            return torch.cat([b[:c] for b, c in zip(bufs, counts)]).to(t.device)
        dist.gather(pad, None, dst=0)
        return None


def _gather_table_to_root(cols, names, counts, rank):
    """One gather for a whole table: the columns `names` of `cols` (1-D tensors of this rank's row count, any dtypes) travel as ONE byte buffer
    (column after column) instead of one collective per column.  Rank 0 gets {name: rank-major concatenation}, the others None."""
    import torch
    n = counts[rank]
    parts = [cols[k].contiguous().view(torch.uint8) for k in names]
    sizes = [cols[k].element_size() for k in names]
    dtypes = [cols[k].dtype for k in names]
    row = sum(sizes)
    dev = cols[names[0]].device
    mine = torch.cat(parts) if n else torch.zeros(0, dtype=torch.uint8, device=dev)
    got = _gather_to_root(mine, [c * row for c in counts], rank)
    if rank != 0:
        return None
    out = {k: [] for k in names}
    base = 0
    for c in counts:
        off = base
        for k, sz, dt in zip(names, sizes, dtypes):
            out[k].append(got[off:off + c * sz].clone().view(dt))          # clone: a fresh, aligned buffer for the typed view
            off += c * sz
        base += c * row
    return {k: torch.cat(v) for k, v in out.items()}


# ---------------------------------------------------------------------------------------------------------------------
# engine adapters: the step below works on torch tensors; an adapter moves them in and out of an engine
# ---------------------------------------------------------------------------------------------------------------------
_TORCH = {"uint8": "uint8", "int32": "int32", "uint64": "int64", "int64": "int64", "float64": "float64"}


def _tdtype(np_dtype):
    import torch
    return getattr(torch, _TORCH[np.dtype(np_dtype).name])


class SvxAdapter(object):
    """libsvx engine, everything resident in HBM (device pointers through the C ABI)."""

    def __init__(self, eng, device):
        self.eng, self.device = eng, device

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def begin_step(self, rank, world):
        """the engine finds its stream start positions inside svx_cluster, over this transport"""
        self.transport = TorchAllGather(world, self.device) if world > 1 else None
        self._at_exchange = False
        self.eng.set_ranks(rank, world, self.transport)

    def end_step(self):
        self.eng.set_ranks(0, 1, None)

    def abort(self):
        """Poison the rank exchange of svx_cluster - ONLY while every peer is known to be entering it: the phases of the step agree on a status
        word before they go on (_agree), so a failure anywhere else is seen by all ranks in the same collective and nobody waits; the one window
        left is between that agreement and svx_cluster itself (which reports its own failures through the exchange)."""
        if self._at_exchange:
            self._at_exchange = False
            self.eng.abort_ranks()

    def stream_end(self):
        return self.eng.stream_positions()[1]

    def collect_counts(self):
        n, nseq, _ = self.eng.collect_counts()
        return n, nseq

    def fetch_signatures(self, with_seq=True):
        """the COLLECT result as device tensors: columns dict, seq_off, seq (with_seq False: the inserted bases stay where they are, seq is empty)"""
        import ctypes as C
        import torch
        from ._lib import _check
        n, nseq = self.collect_counts()
        if not with_seq:
            nseq = 0
        cols = {k: torch.empty(max(1, n), dtype=_tdtype(SIG_DTYPES[k]), device=self.device) for k in SIG_COLS}
        seq_off = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
        seq = torch.zeros(max(1, nseq), dtype=torch.uint8, device=self.device)
        v = _abi.SigView()
        v.on_device, v.n = 1, n
        for k in SIG_COLS:
            setattr(v, k, _abi.ptr(cols[k]))
        v.seq_off, v.seq = _abi.ptr(seq_off), (_abi.ptr(seq) if with_seq else None)
        # torch fills the destination tensors on ITS stream, libsvx copies into them on its own: without this the zero-fill of `seq` can land on top of
        # the copied bases (seen as INS clusters merging on a busy GPU: tests/mp_c3_ranks_one_gpu.py)
        torch.cuda.current_stream().synchronize()
        _check(self.eng.L.svx_collect_fetch(self.eng.ctx, 0, C.byref(v)), "svx_collect_fetch")
        return {k: c[:n] for k, c in cols.items()}, seq_off, seq[:nseq]

    def cluster(self, params, contig_rank, table=None):
        """table None: the resident COLLECT result (source 0); else (cols, seq_off, seq) device tensors (source 2)"""
        import torch
        self._at_exchange = True              # the caller has agreed with every rank that all of them call this now
        if table is None:
            self._at_exchange = False
            self.eng.cluster(params, contig_rank, source=0, fetch=False)
            return
        cols, seq_off, seq = table
        n = int(cols["type"].numel())
        keep = []
        v = _abi.SigView()
        v.on_device, v.n = 1, n
        for k in SIG_COLS:
            c = cols[k].contiguous() if n else torch.zeros(1, dtype=cols[k].dtype, device=self.device)
            keep.append(c)
            setattr(v, k, _abi.ptr(c))
        so = seq_off.contiguous()
        sq = seq.contiguous() if seq.numel() else torch.zeros(1, dtype=torch.uint8, device=self.device)
        keep += [so, sq]
        v.seq_off, v.seq = _abi.ptr(so), _abi.ptr(sq)
        torch.cuda.synchronize()              # the tensors were produced on torch's / RCCL's streams; libsvx runs on its own
        self._at_exchange = False             # from here on svx_cluster tells the others itself when it fails
        self.eng.cluster(params, contig_rank, table=v, source=2, fetch=False)
        self._keep = keep

    def fetch_clusters(self):
        """this rank's cluster table as device tensors (svx_cluster_fetch with device destinations): (cols, members)"""
        import ctypes as C
        import torch
        from ._lib import _check
        eng, dev = self.eng, self.device
        n, nm = C.c_int64(), C.c_int64()
        _check(eng.L.svx_cluster_count(eng.ctx, C.byref(n), C.byref(nm)), "svx_cluster_count")
        n, nm = n.value, nm.value
        cols = {k: torch.empty(max(1, n), dtype=_tdtype(dt), device=dev) for k, dt in CLU_DTYPES.items()}
        member_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        members = torch.empty(max(1, nm), dtype=torch.int32, device=dev)
        cv = _abi.ClusterView()
        cv.n = n
        for k in CLU_DTYPES:
            setattr(cv, k, _abi.ptr(cols[k]))
        cv.member_off, cv.members = _abi.ptr(member_off), _abi.ptr(members)
        torch.cuda.current_stream().synchronize()             # (the allocations above belong to torch's stream: see fetch_signatures)
        _check(eng.L.svx_cluster_fetch(eng.ctx, C.byref(cv)), "svx_cluster_fetch")
        return {k: v[:n] for k, v in cols.items()}, members[:nm]


class HostAdapter(object):
    """An engine with host tables (the oracle in the gloo tests; also libsvx with host arrays): tensors are CPU tensors."""

    def __init__(self, engine, sig_table):
        self.engine, self.sig, self.device = engine, sig_table, "cpu"
        self.ct = None

    def sync(self):
        pass

    def begin_step(self, rank, world):
        self.rank, self.world = rank, world
        self._ends = None

    def end_step(self):
        self.engine.set_chain(None)

    def abort(self):
        pass

    def stream_end(self):
        return self._ends

    def _chain(self, phase, words):
        if phase == 0:
            words[:] = self._starts
        else:
            self._ends = list(words)

    def _exchange_stream_starts(self, tab, params, contig_rank):
        """The checker's version of what svx_cluster does internally: all ranks publish the sizes of their > 100-member partitions per type
        (sorted order), and every rank replays the consumption of the ranks before it with CPython's own generator."""
        import torch.distributed as dist
        sizes = [[] for _ in range(6)]
        if tab.n:
            sidx, pid = self.engine.form_partitions(tab, contig_rank, int(params.partition_max_distance))
            typ = tab.type[:tab.n][sidx]
            cut = np.nonzero(np.diff(pid))[0] + 1
            for a, b in zip(np.concatenate([[0], cut]), np.concatenate([cut, [tab.n]])):
                if b - a > 100:
                    sizes[int(typ[a])].append(int(b - a))
        everyone = [None] * self.world
        dist.all_gather_object(everyone, sizes)
        self._starts = [stream_words_after([n for r in range(self.rank) for n in everyone[r][t]]) for t in range(6)]
        self.engine.set_chain(self._chain)

    def collect_counts(self):
        return self.sig.n, int(self.sig.seq_off[self.sig.n])

    def fetch_signatures(self, with_seq=True):
        import torch
        t = self.sig
        cols = {}
        for k in SIG_COLS:
            a = np.ascontiguousarray(getattr(t, k)[:t.n])
            cols[k] = torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a).clone()
        nseq = int(t.seq_off[t.n])
        return cols, torch.from_numpy(t.seq_off[:t.n + 1].astype(np.int64)), torch.from_numpy(np.ascontiguousarray(t.seq[:nseq])).clone()

    def cluster(self, params, contig_rank, table=None):
        if table is None:
            tab = self.sig
        else:
            cols, seq_off, seq = table
            n = int(cols["type"].numel())
            tab = _abi.SigTable(n, int(seq.numel()))
            for k in SIG_COLS:
                a = cols[k].numpy()
                getattr(tab, k)[:] = a.view(np.uint64) if SIG_DTYPES[k] == np.uint64 else a
            tab.seq_off[:] = seq_off.numpy()
            if seq.numel():
                tab.seq[:seq.numel()] = seq.numpy()
        if getattr(self, "world", 1) > 1:
            self._exchange_stream_starts(tab, params, contig_rank)
        else:
            self._starts = [0] * 6
            self.engine.set_chain(self._chain)
        self.ct = self.engine.cluster(params, contig_rank, table=tab)

    def fetch_clusters(self):
        import torch
        ct = self.ct
        cols = {k: torch.from_numpy(np.ascontiguousarray(getattr(ct, k)[:ct.n])) for k in CLU_DTYPES}
        return cols, torch.from_numpy(np.ascontiguousarray(ct.members[:ct.n_members]))


class StepResult(object):
    """Rank 0's merged result: cluster columns (type-major, then rank-major = the reference's order), member lists as indices
    into the gathered signature table, that table's fixed-width columns (torch tensors), per-rank signature counts."""

    def __init__(self, clusters, member_off, members, sig_cols, sig_counts, chain_end):
        self.clusters, self.member_off, self.members = clusters, member_off, members
        self.sig_cols, self.sig_counts, self.chain_end = sig_cols, sig_counts, chain_end
        self.n = int(clusters["type"].numel()) if clusters is not None else 0

    def set_read_names(self, names_by_rank, stride):
        """names_by_rank[r][i]: the read rank r numbered i; the gathered table carries rank r's ids shifted by r * stride"""
        self.names_by_rank, self.name_stride = names_by_rank, stride

    def read_name(self, read_id):
        """query name behind a read_id of the gathered signature table (sig_cols["read_id"])"""
        r, i = divmod(int(read_id), self.name_stride)
        return self.names_by_rank[r][i]

    def to_host(self):
        out = _abi.ClusterTable(self.n, int(self.members.numel()))
        for k in CLU_DTYPES:
            setattr(out, k, self.clusters[k].cpu().numpy())
        out.member_off = self.member_off.cpu().numpy()
        out.members = self.members.cpu().numpy().astype(np.int32)
        out.n_members = int(self.members.numel())
        out.type_count = [int((out.type == k).sum()) for k in range(6)]
        return out


class RankFailed(RuntimeError):
    """another rank failed in a phase of the step (its own exception is raised there)"""


class _Phase(object):
    """Local work of one phase of the step.  A failure is kept, the rank still takes part in the phase's count exchange (_agree) and all ranks raise
    together right after it - a rank that simply stopped would leave the others waiting in their next collective."""

    def __init__(self):
        self.err = None

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if ev is not None and isinstance(ev, Exception):
            self.err = ev
            return True
        return False


def _agree(phase, values, dev, world):
    """count exchange of a phase + one status word per rank"""
    st = 0 if phase.err is None else 1
    mine = [int(x) for x in values] + [st] if st == 0 else [0] * len(values) + [1]
    got = _all_gather_counts(mine, dev) if world > 1 else [mine]
    if phase.err is not None:
        raise phase.err
    bad = [r for r, c in enumerate(got) if c[-1]]
    if bad:
        raise RankFailed("rank(s) %s failed in this phase of the multi-GPU step" % bad)
    return [c[:-1] for c in got]


def cluster_step(adapter, params, rank, world, contig_gid, contig_rank_global, owner_of_contig, key_base=0, read_base=0,
                 gather_signatures=True, names_of=None, ids_of=None, key_runs=None):
    """One multi-GPU CLUSTER step after this rank's COLLECT.

    contig_gid          int64 [n_local_contig]: global id of every LOCAL contig id the COLLECT tables use
    contig_rank_global  int32 [n_global]: rank of every contig NAME in str order
    owner_of_contig     int32 [n_global] (assign_contigs: whole contigs) or a Windows object (assign_windows: coordinate windows; cuts that are not refined
                        yet are moved into corridors of the collected signatures at the start of the step - one small all-gather)
    key_base            added to the slot half of the emission keys: 2 x (records in file order before this rank's first record)
    read_base           added to the read ids (unique across ranks when reads never span ranks: the synthetic bench layout)
    key_runs            (local_slot_starts, global_slot_starts): a rank that collected several file regions (contig runs of a BAM) numbered
                        its emission slots locally; region k's slots [local_k, local_k+1) start at global_k in file order
                        (harness.collect_bam_sharded exchanges the region sizes).  Replaces key_base.
    names_of, ids_of    real inputs: a read can have records on contigs of different ranks, and the same-read rules of the
                        clustering (SVIM_clustering.py:141-167) need ONE id per read inside a rank's table.  Read ids stay
                        rank-local; foreign rows travel with their read NAMES (names_of(local ids) -> list of str) and the
                        receiving rank interns them into its own numbering (ids_of(list of str) -> ids).
    Returns StepResult on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    dev = adapter.device
    gid = torch.as_tensor(np.asarray(contig_gid), dtype=torch.int64, device=dev)
    windows = owner_of_contig if isinstance(owner_of_contig, Windows) else Windows._whole(np.asarray(owner_of_contig), np.asarray(contig_rank_global, dtype=np.int64), world)
    if key_runs is not None:
        kr_local = torch.as_tensor(np.asarray(key_runs[0], dtype=np.int64), device=dev)
        kr_delta = torch.as_tensor(np.asarray(key_runs[1], dtype=np.int64) - np.asarray(key_runs[0], dtype=np.int64), device=dev)

    def global_keys(key):
        if key_runs is None:
            return key + (int(key_base) << 32)
        if key.numel() == 0:
            return key
        run = (torch.searchsorted(kr_local, key >> 32, right=True) - 1).clamp_min(0)
        return key + (kr_delta[run] << 32)

    adapter.begin_step(rank, world)
    try:
        return _cluster_step(adapter, params, rank, world, contig_gid, contig_rank_global, windows, gather_signatures, names_of, ids_of,
                             read_base, gid, global_keys, dev)
    except BaseException:
        adapter.abort()               # (only does something between the last agreement of the ranks and svx_cluster: see SvxAdapter.abort)
        raise
    finally:
        adapter.end_step()


def _owner_rows(windows, cols, gid=None):
    """owner rank of every row of a signature table (torch columns; gid: local -> global contig ids, None when the columns are global already)"""
    import torch
    c1 = cols["contig"].long()
    c2 = cols["contig2"].long()
    if gid is not None:
        c1 = gid[c1]
        c2 = torch.where(c2 >= 0, gid[c2.clamp_min(0)], c2)
    return windows.owner_of_signatures(cols["type"], c1, c2, cols["start"], cols["end"], cols["pos2"])


def _cluster_step(adapter, params, rank, world, contig_gid, contig_rank_global, windows, gather_signatures, names_of, ids_of, read_base, gid,
                  global_keys, dev):
    import torch
    import torch.distributed as dist
    cols = seq_off = seq = None
    n_own = 0
    # ---- 0. proposed cuts -> corridors ------------------------------------------------------------------------------------
    # (only with assign_windows: every rank publishes the merged [start, end] stretches of its signatures around every cut that lies inside a contig, and all
    # ranks move the cuts into the same corridors - Windows.refine_from)
    if world > 1 and windows.needs_refine():
        max_d = int(params.partition_max_distance)
        radius = max(64 * max_d, 100000)
        mine = np.zeros(0, dtype=np.int64)
        with _Phase() as ph:
            n_own, _ = adapter.collect_counts()
            if n_own:
                cols, seq_off, seq = adapter.fetch_signatures(with_seq=False)
                mine = windows.local_intervals(cols["type"].cpu().numpy(), gid[cols["contig"].long()].cpu().numpy(), cols["start"].cpu().numpy(),
                                               cols["end"].cpu().numpy(), radius)
        cnt0 = _agree(ph, [mine.size], dev, world)
        got = _all_gather_rows(torch.as_tensor(mine, device=dev), [c[0] for c in cnt0])
        windows = windows.refine_from(torch.cat(got).cpu().numpy() if got else mine, max_d, radius)
    # ---- 1. foreign signatures --------------------------------------------------------------------------------------------
    n_foreign = 0
    with _Phase() as ph:
        n_own, _ = adapter.collect_counts()
        if world > 1 and n_own:
            if cols is None:
                cols, seq_off, seq = adapter.fetch_signatures(with_seq=False)    # the columns decide who owns a row; sequences only travel with foreign rows
            foreign = _owner_rows(windows, cols, gid) != rank
            n_foreign = int(foreign.sum().item())
    counts = _agree(ph, [n_foreign], dev, world)
    any_foreign = any(c[0] for c in counts)
    local_rank_arr = np.asarray(contig_rank_global, dtype=np.int32)[np.asarray(contig_gid, dtype=np.int64)]
    if not any_foreign:
        # the common case: every signature stays where it was collected; cluster the resident table as it is (local contig ids)
        adapter.cluster(params, local_rank_arr, table=None)
        n_sig = n_own
        local_to_global_contig = gid
    else:
        # ---- 2a. what this rank sends (local work), agreed sizes, then the rows travel
        fidx = f_cols = f_len = f_seq = lens = mine_names = None
        n_send = f_bytes = 0
        with _Phase() as ph:
            if cols is None:
                cols, seq_off, seq = adapter.fetch_signatures()
                foreign = torch.zeros(0, dtype=torch.bool, device=dev)
            else:
                cols, seq_off, seq = adapter.fetch_signatures()                    # this time with the inserted bases (same rows, same order)
            # globalise ids, split off the foreign rows, exchange them (with their inserted sequences), keep the rows this rank owns
            cols = dict(cols)
            cols["contig"] = gid[cols["contig"].long()].to(torch.int32)
            cols["contig2"] = torch.where(cols["contig2"] >= 0, gid[cols["contig2"].clamp_min(0).long()].to(torch.int32), cols["contig2"])
            cols["read_id"] = cols["read_id"] + read_base
            cols["key"] = global_keys(cols["key"])
            lens = seq_off[1:] - seq_off[:-1]
            fidx = torch.nonzero(foreign).flatten()
            f_cols = {k: cols[k][fidx] for k in SIG_COLS}
            f_len = lens[fidx]
            if names_of is not None:
                mine_names = list(names_of((f_cols["read_id"] - read_base).cpu().numpy()))
            # inserted sequences of the foreign rows (rare: an INS between supplementary segments of another contig)
            f_bytes = int(f_len.sum().item())
            if f_bytes:
                src = torch.repeat_interleave(seq_off[:-1][fidx] - (torch.cumsum(f_len, 0) - f_len), f_len) + torch.arange(f_bytes, device=dev)
                f_seq = seq[src]
            else:
                f_seq = torch.zeros(0, dtype=torch.uint8, device=dev)
            n_send = int(fidx.numel())
        cnt2 = _agree(ph, [n_send, f_bytes], dev, world)
        g_names = None
        if names_of is not None:
            g_names = [None] * world
            dist.all_gather_object(g_names, mine_names)
        rows = [c[0] for c in cnt2]
        # two collectives: the fixed-width columns + the sequence lengths as ONE byte buffer (round 5; twelve all-gathers until then), the inserted bases
        tab = dict(f_cols)
        tab["__len"] = f_len
        g_tab = _all_gather_table(tab, list(SIG_COLS) + ["__len"], rows)
        g_len = g_tab.pop("__len")
        g_cols = g_tab
        g_seq = _all_gather_rows(f_seq, [c[1] for c in cnt2])
        # ---- 2b. this rank's table: its own rows + the rows it received (local work), then all ranks enter svx_cluster together
        with _Phase() as ph:
            keep = ~foreign
            parts_cols = {k: [cols[k][keep]] for k in SIG_COLS}
            parts_len, parts_seq = [lens[keep]], []
            kidx = torch.nonzero(keep).flatten()
            k_bytes = int(lens[keep].sum().item())
            if k_bytes:
                kl = lens[kidx]
                src = torch.repeat_interleave(seq_off[:-1][kidx] - (torch.cumsum(kl, 0) - kl), kl) + torch.arange(k_bytes, device=dev)
                parts_seq.append(seq[src])
            for r in range(world):
                if r == rank or rows[r] == 0:
                    continue
                take = _owner_rows(windows, {k: g_cols[k][r] for k in ("type", "contig", "contig2", "start", "end", "pos2")}) == rank
                if not bool(take.any()):
                    continue
                tidx = torch.nonzero(take).flatten()
                for k in SIG_COLS:
                    if k == "read_id" and g_names is not None:
                        got = [g_names[r][int(i)] for i in tidx.tolist()]
                        parts_cols[k].append(torch.as_tensor(np.asarray(ids_of(got), dtype=np.int32), device=dev) + read_base)
                    else:
                        parts_cols[k].append(g_cols[k][r][tidx])
                tl = g_len[r][tidx]
                parts_len.append(tl)
                tb = int(tl.sum().item())
                if tb:
                    starts = torch.cumsum(g_len[r], 0) - g_len[r]
                    src = torch.repeat_interleave(starts[tidx] - (torch.cumsum(tl, 0) - tl), tl) + torch.arange(tb, device=dev)
                    parts_seq.append(g_seq[r][src])
            cols = {k: torch.cat(v) for k, v in parts_cols.items()}
            lens = torch.cat(parts_len)
            seq = torch.cat(parts_seq) if parts_seq else torch.zeros(0, dtype=torch.uint8, device=dev)
            # list order = emission order (the partition sort is stable with respect to it): sort by the global key
            order = torch.sort(cols["key"], stable=True).indices
            starts = torch.cumsum(lens, 0) - lens
            cols = {k: v[order] for k, v in cols.items()}
            ol = lens[order]
            tot = int(ol.sum().item())
            if tot:
                src = torch.repeat_interleave(starts[order] - (torch.cumsum(ol, 0) - ol), ol) + torch.arange(tot, device=dev)
                seq = seq[src]
            seq_off = torch.zeros(ol.numel() + 1, dtype=torch.int64, device=dev)
            torch.cumsum(ol, 0, out=seq_off[1:])
            n_sig = int(ol.numel())
        _agree(ph, [], dev, world)
        adapter.cluster(params, np.asarray(contig_rank_global, dtype=np.int32), table=(cols, seq_off, seq))
        local_to_global_contig = None
    # ---- 3. final candidate gather to rank 0 ----------------------------------------------------------------------------
    ncl = nmem = 0
    with _Phase() as ph:
        if local_to_global_contig is not None:                                # the resident table was clustered: its columns, globalised for the gather
            if gather_signatures and cols is None and n_own:
                cols, seq_off, seq = adapter.fetch_signatures(with_seq=False)
            if cols is not None:
                cols = dict(cols)
                cols["contig"] = gid[cols["contig"].long()].to(torch.int32)
                cols["contig2"] = torch.where(cols["contig2"] >= 0, gid[cols["contig2"].clamp_min(0).long()].to(torch.int32), cols["contig2"])
                cols["read_id"] = cols["read_id"] + read_base
                cols["key"] = global_keys(cols["key"])
        c_cols, members = adapter.fetch_clusters()
        if local_to_global_contig is not None and int(c_cols["type"].numel()):
            c_cols = dict(c_cols)
            c_cols["contig"] = local_to_global_contig[c_cols["contig"].long()].to(torch.int32)
            c_cols["contig2"] = torch.where(c_cols["contig2"] >= 0, local_to_global_contig[c_cols["contig2"].clamp_min(0).long()].to(torch.int32), c_cols["contig2"])
        ncl, nmem = int(c_cols["type"].numel()), int(members.numel())
    allc = _agree(ph, [ncl, nmem, n_sig], dev, world)
    sig_counts = [c[2] for c in allc]
    sig_base = int(sum(sig_counts[:rank]))
    members = members.to(torch.int64) + sig_base                           # indices into the rank-major gathered signature table
    if world == 1:
        g = {k: v for k, v in c_cols.items()}
        g_mem = members
        g_sig = cols if gather_signatures else None
    else:
        # three collectives: cluster records, member lists, signature columns (each table as one byte buffer)
        g = _gather_table_to_root(c_cols, list(CLU_DTYPES), [c[0] for c in allc], rank)
        g_mem = _gather_to_root(members, [c[1] for c in allc], rank)
        g_sig = None
        if gather_signatures:
            empty = {k: torch.zeros(0, dtype=_tdtype(SIG_DTYPES[k]), device=dev) for k in SIG_COLS}
            src = cols if cols is not None else empty
            g_sig = _gather_table_to_root(src, list(SIG_COLS), sig_counts, rank)
    chain_end = adapter.stream_end()
    if rank != 0:
        return None
    # rank-major concatenation -> type-major: a stable sort by type keeps, inside a type, rank order and each rank's own order -
    # which is the reference's (unilocal: (contig name, (start + end) / 2); bilocal: partition order)
    sizes = g["size"].to(torch.int64)
    src_off = torch.cumsum(sizes, 0) - sizes
    order = torch.sort(g["type"].to(torch.int64), stable=True).indices
    out_cols = {k: v[order] for k, v in g.items()}
    out_sizes = sizes[order]
    n = int(out_sizes.numel())
    member_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(out_sizes, 0, out=member_off[1:])
    nm = int(g_mem.numel())
    src_idx = torch.repeat_interleave(src_off[order] - member_off[:-1], out_sizes) + torch.arange(nm, dtype=torch.int64, device=dev)
    res = StepResult(out_cols, member_off, g_mem[src_idx], g_sig, sig_counts, chain_end)
    res.windows = windows                                                  # the (refined) ownership every rank used
    return res
