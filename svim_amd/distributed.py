"""Table helpers shared by the multi-batch COLLECT driver and the partition-sharded clustering modes (svx_cluster_set_shard /
_by_origin); the multi-GPU step itself lives in svim_amd/multigpu.py (contig-sharded ranks, every partition local).

Sharding (DESIGN.md "Multi-GPU"):
  * COLLECT shards the record batch by contiguous record ranges - records are independent;
  * the per-rank signature tables are all-gathered (rank-major = file order) so that every rank sees the
    complete list: sort + partition + the sequential random.sample stream (which is carried across partitions
    of a type) are recomputed redundantly - they are cheap - and each rank clusters only the partitions with
    index % world == rank (the quadratic work);
  * one gather of the fixed-width cluster records to rank 0, merged by partition index.
"""
import numpy as np

from . import _abi
from ._abi import CLU_DTYPES, ClusterTable, SIG_DTYPES, SigTable


def merge_cluster_tables(parts, contig_rank):
    """Merge per-shard ClusterTables (each carrying part_index) into the single-GPU result: clusters ordered by
    type, then - unilocal types - by (contig name rank, start+end) with ties in partition order, - bilocal types -
    in partition order (src/svim/SVIM_clustering.py:381,383)."""
    parts = [p for p in parts if p.n > 0]
    n = sum(p.n for p in parts)
    nm = sum(p.n_members for p in parts)
    out = ClusterTable(n, nm)
    if n == 0:
        return out
    cat = {k: np.concatenate([getattr(p, k)[:p.n] for p in parts]) for k in CLU_DTYPES}
    part_index = np.concatenate([p.part_index[:p.n] for p in parts])
    src_off, base = [], 0
    for p in parts:
        src_off.append(p.member_off[:p.n] + base)
        base += p.n_members
    src_off = np.concatenate(src_off)
    members = np.concatenate([p.members[:p.n_members] for p in parts])
    rank = np.asarray(contig_rank)
    t = cat["type"].astype(np.int64)
    uni = t <= 2
    k1 = np.where(uni, rank[cat["contig"]].astype(np.int64), 0)
    k2 = np.where(uni, cat["start"].astype(np.int64) + cat["end"].astype(np.int64), 0)
    # within one shard the device order is already final; across shards ties are broken by partition index, and
    # clusters of one partition always come from the same shard in label order (kept by the stable sort)
    order = np.lexsort((np.arange(n), part_index, k2, k1, t))
    for k in CLU_DTYPES:
        setattr(out, k, cat[k][order])
    out.part_index = part_index[order]
    sizes = out.size.astype(np.int64)
    out.member_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sizes, out=out.member_off[1:])
    so = src_off[order]
    # member j of output cluster i comes from members[so[i] + j]: one vectorised gather
    src_idx = np.repeat(so - out.member_off[:-1], sizes) + np.arange(nm, dtype=np.int64)
    out.members = members[src_idx].astype(np.int32) if nm else np.zeros(0, dtype=np.int32)
    out.n, out.n_members = n, nm
    out.type_count = [int((out.type == k).sum()) for k in range(6)]
    return out


def concat_sig_tables(tables):
    """Rank-major concatenation of signature tables (= file order when records were sharded contiguously)."""
    n = sum(t.n for t in tables)
    nseq = sum(int(t.seq_off[t.n]) for t in tables)
    out = SigTable(n, nseq)
    pos, spos = 0, 0
    for t in tables:
        for k in SIG_DTYPES:
            getattr(out, k)[pos:pos + t.n] = getattr(t, k)[:t.n]
        m = int(t.seq_off[t.n])
        out.seq_off[pos:pos + t.n + 1] = t.seq_off[:t.n + 1] + spos
        out.seq[spos:spos + m] = t.seq[:m]
        pos += t.n
        spos += m
    out.key[:] = np.arange(n, dtype=np.uint64)       # list position is the order from here on
    return out


def _pack_sig(t):
    cols = [getattr(t, k)[:t.n].view(np.uint8).reshape(-1) for k in SIG_DTYPES]
    head = np.array([t.n, int(t.seq_off[t.n])], dtype=np.int64).view(np.uint8)
    return np.concatenate([head] + cols + [t.seq_off[:t.n + 1].view(np.uint8), t.seq[:int(t.seq_off[t.n])]])


def _unpack_sig(buf):
    n, nseq = (int(x) for x in buf[:16].view(np.int64))
    t = SigTable(n, nseq)
    p = 16
    for k, dt in SIG_DTYPES.items():
        nb = n * np.dtype(dt).itemsize
        setattr(t, k, buf[p:p + nb].view(dt).copy())
        p += nb
    t.seq_off = buf[p:p + (n + 1) * 8].view(np.int64).copy()
    p += (n + 1) * 8
    t.seq = buf[p:p + nseq].copy() if nseq else np.zeros(1, dtype=np.uint8)
    return t


def _pack_clu(c):
    head = np.array([c.n, c.n_members], dtype=np.int64).view(np.uint8)
    cols = [getattr(c, k)[:c.n].view(np.uint8).reshape(-1) for k in CLU_DTYPES]
    return np.concatenate([head] + cols + [c.member_off[:c.n + 1].view(np.uint8), c.members[:c.n_members].view(np.uint8),
                                           np.asarray(c.part_index[:c.n], dtype=np.int64).view(np.uint8)])


def _unpack_clu(buf):
    n, nm = (int(x) for x in buf[:16].view(np.int64))
    c = ClusterTable(n, nm)
    p = 16
    for k, dt in CLU_DTYPES.items():
        nb = n * np.dtype(dt).itemsize
        setattr(c, k, buf[p:p + nb].view(dt).copy())
        p += nb
    c.member_off = buf[p:p + (n + 1) * 8].view(np.int64).copy()
    p += (n + 1) * 8
    c.members = buf[p:p + nm * 4].view(np.int32).copy()
    p += nm * 4
    c.part_index = buf[p:p + n * 8].view(np.int64).copy()
    return c


def all_gather_bytes(arr, device=None):
    """all_gather of variable-length uint8 arrays through torch.distributed (padded to the max length)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    n = torch.tensor([arr.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes + [1])
    buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if arr.size:
        buf[:arr.size] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    outs = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]


def all_gather_signatures(table, device=None):
    """Every rank ends up with the rank-major concatenation of all ranks' signature tables."""
    return concat_sig_tables([_unpack_sig(b) for b in all_gather_bytes(_pack_sig(table), device)])


def gather_clusters(ct, contig_rank, device=None):
    """Final candidate gather: merged ClusterTable (identical on every rank; rank 0 is the consumer)."""
    return merge_cluster_tables([_unpack_clu(b) for b in all_gather_bytes(_pack_clu(ct), device)], contig_rank)


# ---------------------------------------------------------------------------------------------------------------------
# device-resident multi-GPU step (bench.py --gpus N): everything stays in HBM, the exchange is RCCL over xGMI
# ---------------------------------------------------------------------------------------------------------------------
_DEV_COLS = (("type", "uint8"), ("src", "uint8"), ("aux", "uint8"), ("contig", "int32"), ("start", "int32"), ("end", "int32"),
             ("contig2", "int32"), ("pos2", "int32"), ("read_id", "int32"))


def _all_gather_var(t, counts, dist, torch):
    """all-gather of 1-D device tensors with per-rank lengths `counts` -> concatenation in rank order."""
    mx = max(max(counts), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = torch.empty(mx * len(counts), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + c] for r, c in enumerate(counts)])


class DeviceClusterTable:
    """Merged cluster table resident in HBM (torch tensors, same columns as ClusterTable); to_host() for consumers on the CPU."""

    def __init__(self, cols, member_off, members, part_index):
        self.cols, self.member_off, self.members, self.part_index = cols, member_off, members, part_index
        self.n, self.n_members = int(part_index.numel()), int(members.numel())

    def to_host(self):
        out = ClusterTable(self.n, self.n_members)
        for k in CLU_DTYPES:
            setattr(out, k, self.cols[k].cpu().numpy())
        out.member_off = self.member_off.cpu().numpy()
        out.members = self.members.cpu().numpy()
        out.part_index = self.part_index.cpu().numpy()
        out.n, out.n_members = self.n, self.n_members
        out.type_count = [int((out.type == k).sum()) for k in range(6)]
        return out


def fetch_clusters_device(eng, dev):
    """This rank's cluster table as device tensors (svx_cluster_fetch with device destinations): (cols, members, part_index)."""
    import ctypes as C
    import torch
    from ._lib import _check
    n, nm = C.c_int64(), C.c_int64()
    _check(eng.L.svx_cluster_count(eng.ctx, C.byref(n), C.byref(nm)), "svx_cluster_count")
    n, nm = n.value, nm.value
    tdt = {np.uint8: torch.uint8, np.int32: torch.int32, np.float64: torch.float64}
    cols = {k: torch.empty(max(1, n), dtype=tdt[dt], device=dev) for k, dt in CLU_DTYPES.items()}
    member_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    members = torch.empty(max(1, nm), dtype=torch.int32, device=dev)
    part_index = torch.empty(max(1, n), dtype=torch.int64, device=dev)
    cv = _abi.ClusterView()
    cv.n = n
    for k in CLU_DTYPES:
        setattr(cv, k, _abi.ptr(cols[k]))
    cv.member_off, cv.members = _abi.ptr(member_off), _abi.ptr(members)
    _check(eng.L.svx_cluster_fetch(eng.ctx, C.byref(cv)), "svx_cluster_fetch")
    _check(eng.L.svx_cluster_fetch_part_index(eng.ctx, _abi.ptr(part_index)), "svx_cluster_fetch_part_index")
    return {k: v[:n] for k, v in cols.items()}, members[:nm], part_index[:n]


def gather_clusters_device(eng, contig_rank, dev):
    """RCCL all-gather of the per-rank cluster tables and the merge of merge_cluster_tables, all on the device: nothing of the
    exchange touches the host (the member lists are ~4 B per signature; a host merge grows with the world size)."""
    import torch
    import torch.distributed as dist
    cols, members, part_index = fetch_clusters_device(eng, dev)
    world = dist.get_world_size()
    cnt = torch.tensor([part_index.numel(), members.numel()], dtype=torch.int64, device=dev)
    allc = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, cnt)
    ns = [int(c[0].item()) for c in allc]
    nms = [int(c[1].item()) for c in allc]
    g = {k: _all_gather_var(v, ns, dist, torch) for k, v in cols.items()}
    g_part = _all_gather_var(part_index, ns, dist, torch)
    g_mem = _all_gather_var(members, nms, dist, torch)
    return merge_gathered_clusters(g, g_part, g_mem, contig_rank)


def merge_gathered_clusters(g, g_part, g_mem, contig_rank):
    """The merge of merge_cluster_tables on torch tensors (any device): `g` = cluster columns of all ranks concatenated rank-major,
    `g_part` their global partition indices, `g_mem` the member lists concatenated in the same order."""
    import torch
    dev = g_part.device
    n, nm = int(g_part.numel()), int(g_mem.numel())
    sizes = g["size"].to(torch.int64)
    # first member of every gathered cluster inside g_mem (tables are concatenated rank-major, members likewise)
    src_off = torch.cumsum(sizes, 0) - sizes
    rank_t = torch.as_tensor(np.asarray(contig_rank), dtype=torch.int64, device=dev)
    t = g["type"].to(torch.int64)
    uni = t <= 2
    zero = torch.zeros_like(t)
    k1 = torch.where(uni, rank_t[g["contig"].to(torch.int64).clamp_min(0)], zero)
    k2 = torch.where(uni, g["start"].to(torch.int64) + g["end"].to(torch.int64), zero)
    # lexsort((arange, part_index, k2, k1, t)) as a chain of stable sorts, least significant key first
    order = torch.arange(n, dtype=torch.int64, device=dev)
    for key in (g_part, k2, k1, t):
        order = order[torch.sort(key[order], stable=True).indices]
    out_cols = {k: v[order] for k, v in g.items()}
    out_sizes = sizes[order]
    member_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(out_sizes, 0, out=member_off[1:])
    src_idx = torch.repeat_interleave(src_off[order] - member_off[:-1], out_sizes) + torch.arange(nm, dtype=torch.int64, device=dev)
    return DeviceClusterTable(out_cols, member_off, g_mem[src_idx], g_part[order])
