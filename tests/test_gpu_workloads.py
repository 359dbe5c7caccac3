"""GPU parity at workload level (`-m gpu`): every BASELINE.json configuration that fits one GPU, generated on the device from its
seed, through the C ABI (svx_collect + svx_cluster) against the oracle on the same records - signature tables and cluster tables
bit-identical (FP columns within 1e-12 of the oracle; the oracle itself is pinned to the reference by tests/golden).

    configs[1]  ONT profile, DEL/INS/INV                                   250 k reads (a quarter of the bench size), ~90 s of oracle
    configs[2]  HiFi profile, full SV-type set (stand-in, svim_amd/workloads.py c2)
    configs[4]  60x CLR profile, --partition_max_distance in {1000, 5000, 20000, 100000} (stand-in, c4)
"""
import collections
import os
import types

import numpy as np
import pytest

from svim_amd import _abi

pytestmark = pytest.mark.gpu


def _options(pmd=1000, all_bnds=False):
    return types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                                 partition_max_distance=pmd, position_distance_normalizer=900, edit_distance_normalizer=1.0,
                                 cluster_max_distance=0.5, all_bnds=all_bnds)


@pytest.fixture(scope="module")
def eng():
    from svim_amd import _lib
    return _lib.Engine(0)


def _both(eng, oracle, batch, g_off, genome, o):
    import torch
    p = _abi.Params.from_options(o)
    eng.set_genome(g_off, genome, on_device=True)
    rank = batch.t["contig_rank"].cpu().numpy().astype(np.int32)
    sig, bnd = eng.collect(batch.struct(), p)
    ct = eng.cluster(p, rank, source=0)
    st = eng.stats()
    torch.cuda.synchronize()
    oracle.set_genome(g_off.cpu().numpy().astype(np.int64), genome.cpu().numpy())
    hb = batch.slice_records(0, batch.n_rec)
    osig, obnd = oracle.collect(hb, p)
    oct_ = oracle.cluster(p, hb.contig_rank, source=0)
    assert sig.first_difference(osig) is None
    assert bnd.first_difference(obnd) is None
    assert ct.first_difference(oct_, rtol=1e-12) is None
    return sig, ct, st


def test_c2_hifi_full_sv_type_set_vs_oracle(eng, oracle):
    from svim_amd import workloads
    prof = workloads.profile("c2", 0.02)
    batch, genome, g_off, meta = workloads.make_batch_full(prof, seed=3, device="cuda:0")
    assert all(v > 0 for v in meta["reads_by_layout"].values())
    sig, ct, st = _both(eng, oracle, batch, g_off, genome, _options(all_bnds=True))
    by = collections.Counter(zip(sig.type.tolist(), sig.src.tolist()))
    # every signature type, CIGAR- and split-read-derived DEL / INS both present
    for key in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 1), (3, 1), (4, 1), (5, 1)):
        assert by[key] > 0, (key, by)
    assert all(c > 0 for c in ct.type_count), ct.type_count
    assert st["n_edit_pairs"] > 1000


@pytest.mark.parametrize("pmd", [1000, 5000, 20000, 100000])
def test_c4_clr_partition_max_distance_sweep_vs_oracle(eng, oracle, pmd):
    from svim_amd import workloads
    prof = workloads.profile("c4", 0.01)
    batch, genome, g_off, meta = workloads.make_batch_full(prof, seed=3, device="cuda:0")
    sig, ct, st = _both(eng, oracle, batch, g_off, genome, _options(pmd=pmd))
    assert st["n_large_partitions"] >= 2          # beyond 100 members: random.sample pool path; at 100000 beyond 1045: set path


@pytest.mark.skipif(os.environ.get("SVX_SKIP_SLOW") == "1", reason="SVX_SKIP_SLOW=1")
def test_c1_ont_quarter_scale_vs_oracle(eng, oracle):
    """configs[1] at 250 k reads / 62.5 Mb (same coverage and site density as the 1 M-read bench batch)."""
    import torch
    from svim_amd import devsynth
    batch, genome, meta = devsynth.make_batch(n_reads=250_000, contig_len=62_500_000, seed=2, device="cuda:0")
    g_off = torch.tensor([0, genome.numel()], dtype=torch.int64, device="cuda:0")
    sig, ct, st = _both(eng, oracle, batch, g_off, genome, _options())
    assert sig.n > 150_000 and ct.n > 5_000 and st["n_large_partitions"] > 100
