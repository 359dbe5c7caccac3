"""Worker of tests/test_gpu_parity.py::test_two_ranks_share_one_gpu_over_gloo (run under torch.distributed.run, 2 ranks, both on cuda:0, backend gloo -
RCCL refuses two ranks on one device): the contig-sharded step of svim_amd/multigpu.py with the REAL engine and DEVICE tensors at world size 2 -
foreign BND rows cross ranks, svx_cluster's rank exchange runs over the process group, rank 0 gathers - against the oracle on the union of both ranks'
inputs.  Prints one line: TWO_RANKS_OK <clusters> <foreign rows> or the first difference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = "cuda:0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svim_amd import _abi, _lib, multigpu as MG, workloads
    import types
    o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                              partition_max_distance=5000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5,
                              all_bnds=False)
    p = _abi.Params.from_options(o)
    prof = workloads.profile("c2", 0.012)

    def make(r):
        """rank r's batch with GLOBAL contig ids (rank-major names g<r>_<name>), half of its cross-contig segment rows pointing at the other rank"""
        b, genome, g_off, meta = workloads.make_batch_full(prof, seed=5 + r, device=dev)
        n_local = b.n_contig
        base = r * n_local
        gen = torch.Generator(device=dev)
        gen.manual_seed(77 + r)
        counts = (b.t["seg_off"][1:] - b.t["seg_off"][:-1]).to(torch.int64)
        prim_tid = torch.repeat_interleave(b.t["tid"].to(torch.int64), counts)
        st_local = b.t["seg_tid"][:b.n_seg].to(torch.int64)
        cross = (st_local != prim_tid) & (torch.rand(b.n_seg, generator=gen, device=dev) < 0.5)
        other = (r + 1) % world
        b.t["seg_tid"][:b.n_seg] = torch.where(cross, other * n_local + st_local, base + st_local).to(b.t["seg_tid"].dtype)
        b.t["tid"] = b.t["tid"] + base
        names = ["g%03d_%s" % (q, nm) for q in range(world) for nm in b.references]
        order = sorted(range(len(names)), key=lambda i: names[i])
        crank = np.zeros(len(names), dtype=np.int32)
        crank[order] = np.arange(len(names), dtype=np.int32)
        b.t["contig_rank"] = torch.as_tensor(crank, device=dev)
        b.n_contig = len(names)
        g_off_global = torch.zeros(len(names) + 1, dtype=torch.int64, device=dev)
        g_off_global[base:base + n_local + 1] = g_off
        g_off_global[base + n_local + 1:] = g_off[-1]
        return b, genome, g_off_global, crank, n_local, int(cross.sum().item())

    b, genome, g_off_global, crank, n_local, n_cross = make(rank)
    owner = np.repeat(np.arange(world, dtype=np.int32), n_local)
    eng = _lib.Engine(0)
    eng.set_genome(g_off_global, genome, on_device=True)
    sizes = MG._all_gather_counts([b.n_rec, int(b.t["read_id"].max().item()) + 1], dev)
    key_base = 2 * sum(c[0] for c in sizes[:rank])
    read_base = sum(c[1] for c in sizes[:rank])
    ad = MG.SvxAdapter(eng, dev)
    gid = np.arange(len(crank), dtype=np.int64)
    verdict = "ok"
    for it in range(2):                                    # twice: the second step runs with warm buffers
        eng.collect(b.struct(), p, fetch=False)
        res = MG.cluster_step(ad, p, rank, world, gid, crank, owner, key_base=key_base, read_base=read_base)
    if rank == 0:
        from oracle import oracle as om
        orc = om.Oracle()
        tabs = []
        glen = 0
        parts_g, offs = [], [0]
        kb = rb = 0
        for r in range(world):
            br, gr, goff_r, _, _, _ = (b, genome, g_off_global, None, None, None) if r == rank else make(r)
            hb = br.slice_records(0, br.n_rec)
            hb.arrays["order"] = (hb.arrays["order"].astype(np.int64) + kb).astype(np.uint32)
            hb.arrays["seg_order"] = (hb.arrays["seg_order"].astype(np.int64) + kb).astype(np.uint32)
            hb.arrays["read_id"] = (hb.arrays["read_id"].astype(np.int64) + rb).astype(np.int32)
            sig, _ = orc.collect(hb, p)
            tabs.append(sig)
            kb += 2 * br.n_rec
            rb += int(br.t["read_id"].max().item()) + 1
            parts_g.append(gr.cpu().numpy())
        # one genome with every rank's contigs
        total = np.concatenate(parts_g)
        goff = np.zeros(len(crank) + 1, dtype=np.int64)
        at = 0
        for r in range(world):
            _, _, goff_r, _, _, _ = (b, genome, g_off_global, None, None, None) if r == rank else make(r)
            lo = goff_r.cpu().numpy()[r * n_local:(r + 1) * n_local + 1]
            goff[r * n_local:(r + 1) * n_local + 1] = at + (lo - lo[0])
            at += int(lo[-1] - lo[0])
        goff[world * n_local:] = at
        orc.set_genome(goff, total)
        n = sum(t.n for t in tabs)
        nseq = sum(int(t.seq_off[t.n]) for t in tabs)
        full_tab = _abi.SigTable(n, nseq)
        a = s = 0
        for t in tabs:
            for k in _abi.SIG_DTYPES:
                getattr(full_tab, k)[a:a + t.n] = getattr(t, k)[:t.n]
            m = int(t.seq_off[t.n])
            full_tab.seq_off[a:a + t.n + 1] = t.seq_off[:t.n + 1] + s
            full_tab.seq[s:s + m] = t.seq[:m]
            a += t.n
            s += m
        full = orc.cluster(p, crank, table=full_tab)
        got = res.to_host()
        if got.n != full.n or list(got.type_count) != list(full.type_count):
            verdict = "n %d/%d type_count %r/%r" % (got.n, full.n, got.type_count, full.type_count)
        else:
            for k in _abi.CLU_DTYPES:
                x, y = getattr(got, k), getattr(full, k)[:full.n]
                same = (np.isnan(x) & np.isnan(y)) | (np.abs(x - y) <= 1e-12 * np.maximum(1.0, np.abs(y))) if x.dtype == np.float64 else x == y
                if not same.all():
                    i = int(np.nonzero(~same)[0][0])
                    verdict = "%s[%d]: %r != %r" % (k, i, x[i], y[i])
                    break
            if verdict == "ok":
                keys = res.sig_cols["key"].cpu().numpy()
                if not np.array_equal(keys[got.members], full_tab.key[:n].astype(np.int64)[full.members[:full.n_members]]):
                    verdict = "member keys differ"
        n_foreign = 0
        for r, t in enumerate(tabs):
            oc = MG.owner_contig(t.type[:t.n], t.contig[:t.n], t.contig2[:t.n])
            n_foreign += int((owner[oc] != r).sum())
        print("TWO_RANKS_OK %d %d %d" % (got.n, n_cross, n_foreign) if verdict == "ok" else "TWO_RANKS_FAIL " + verdict, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
