"""The GPU's second DEFLATE decoder - one BGZF block per LANE (svim_amd/csrc/inflate_lanes.hpp) - built for the host, where every lane is just a serial
decoder (tools/inflate_lanes_host_test.cpp), and checked against zlib: random buffers at every level / strategy, the BGZF blocks of BAM files with and
without base qualities, damaged streams.  A lane may GIVE UP a block (stored blocks, code tables beyond its share of LDS, anything irregular: the GPU
then redoes the block with the wave-per-block decoder of inflate_core.hpp) - what it answers must be zlib's bytes."""
import os
import subprocess

import numpy as np
import pytest

from svim_amd import records, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(REPO, "tools", "inflate_lanes_host_test.cpp")]
INC = ["-I", os.path.join(REPO, "svim_amd", "csrc")]


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("inflate_lanes") / "inflate_lanes_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", *INC, *SRC, "-lz", "-o", out])
    return out


@pytest.mark.parametrize("depth", [2, 6])
def test_lane_decoder_at_other_pipeline_depths(tmp_path, depth):
    """INFL_DEPTH = trips between the request of a match chunk and its store (3 in the shipped build).  The rules that keep a request from reading bytes whose
    store is still pending depend on it; depth 6 is where a wrong rule shows within a few hundred BAM blocks (it found one: a pending chunk can lie far behind
    the output position when a short-distance match waited for empty slots)."""
    out = str(tmp_path / "inflate_lanes_depth")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DINFL_DEPTH=%d" % depth, *INC, *SRC, "-lz", "-o", out])
    path = str(tmp_path / "t.bam")
    _bam_with_qualities(path, False)
    run = subprocess.run([out, path], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and " 0 mismatches" in run.stdout.splitlines()[0], (run.stdout[-400:], run.stderr[-1000:])
    run = subprocess.run([out, "--fuzz", "600"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "600 buffers, 0 mismatches" in run.stdout, (run.stdout[-400:], run.stderr[-1000:])


def test_lane_decoder_fuzz_vs_zlib(host_binary):
    out = subprocess.run([host_binary, "--fuzz", "1500"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "1500 buffers, 0 mismatches" in out.stdout
    decoded = int(out.stdout.split("mismatches,")[1].split("decoded")[0])
    assert decoded > 400, out.stdout                    # the rest: level-0 / incompressible buffers (stored blocks) and random bytes whose code tables exceed a lane's LDS


def _bam_with_qualities(path, with_qual):
    contigs = [("chr1", 200000)]
    refs = synth.make_reference(3, contigs)
    recs = synth.planted_reads(5, 600, refs, ["chr1"], [200000], n_sites=30, types=("DEL", "INS", "INV"))
    if with_qual:
        rng = np.random.default_rng(11)
        for r in recs:
            n = len(r.query_sequence) if getattr(r, "query_sequence", None) else 0
            if n:
                r.query_qualities = np.clip(rng.normal(18, 8, n), 1, 50).astype(np.uint8).tolist()
    records.write_bam(path, ["chr1"], [200000], synth.coordinate_sort(recs))


@pytest.mark.parametrize("with_qual", [False, True])
def test_lane_decoder_bam_blocks_vs_zlib(host_binary, tmp_path, with_qual):
    path = str(tmp_path / "t.bam")
    try:
        _bam_with_qualities(path, with_qual)
    except (AttributeError, TypeError):
        if not with_qual:
            raise
        pytest.skip("the test records carry no settable qualities")
    out = subprocess.run([host_binary, path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    head = out.stdout.splitlines()[0]
    assert " 0 mismatches" in head, out.stdout
    blocks = int(head.split()[0])
    gave_up = int(head.split("mismatches,")[1].split("given up")[0])
    assert blocks > 3 and gave_up * 4 <= blocks, head         # BAM blocks are what the table budget was sized for


def test_lane_decoder_damaged_streams_are_refused_or_answered_like_zlib(tmp_path):
    """Bit flips, overwritten headers, truncation, a lying output size: a lane ends in a refusal or in exactly what zlib accepts - under AddressSanitizer +
    UndefinedBehaviorSanitizer, within a bounded number of trips, without a write outside its output or its scratch (on the GPU such a write would corrupt
    another lane's tables or the next block's output)."""
    out = str(tmp_path / "inflate_lanes_asan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                            *INC, *SRC, "-lz", "-o", out], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([out, "--damaged", "100"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "600 streams, 0 bad" in run.stdout
    run = subprocess.run([out, "--fuzz", "60"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and "60 buffers, 0 mismatches" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
