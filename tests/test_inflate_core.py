"""The GPU's DEFLATE decoder (svim_amd/csrc/inflate_core.hpp) built for the host with its lane operations emulated (tools/inflate_host_test.cpp)
and checked against zlib: random buffers at every level / strategy (stored, fixed and dynamic blocks) and the BGZF blocks of a BAM file."""
import os
import subprocess

import pytest

from svim_amd import records, synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("inflate") / "inflate_host_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DINF_HOST", "-I", os.path.join(REPO, "svim_amd", "csrc"),
                           os.path.join(REPO, "tools", "inflate_host_test.cpp"), "-lz", "-o", out])
    return out


def test_inflate_core_fuzz_vs_zlib(host_binary):
    out = subprocess.run([host_binary, "--fuzz", "1500"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "1500 buffers, 0 mismatches" in out.stdout
    # the multi-window steps of literal-heavy streams (base qualities) are part of what was checked, with 2, 3 and 4 windows
    line = out.stdout.split("multi-window steps:")[1].splitlines()[0]
    taken = int(line.split()[0])
    by_width = [int(x) for x in line.split("windows:")[1].split(")")[0].replace("/", " ").split()]
    assert taken > 100000 and all(n > 1000 for n in by_width), out.stdout[-300:]


def test_inflate_core_bam_blocks_vs_zlib(host_binary, tmp_path):
    contigs = [("chr1", 200000)]
    refs = synth.make_reference(3, contigs)
    recs = synth.planted_reads(5, 600, refs, ["chr1"], [200000], n_sites=30, types=("DEL", "INS", "INV"))
    path = str(tmp_path / "t.bam")
    records.write_bam(path, ["chr1"], [200000], synth.coordinate_sort(recs))
    out = subprocess.run([host_binary, path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert " 0 mismatches" in out.stdout


def test_inflate_core_damaged_streams_stay_inside_their_buffers(tmp_path):
    """Bit flips, overwritten headers, truncation and a lying output size: the decoder ends in an error code or in some output, never in an access outside
    its buffers (the host build under AddressSanitizer + UndefinedBehaviorSanitizer: array bounds of the LDS scratch members included) and never writes
    behind the output capacity.  On the GPU such an access would take the whole process down (a damaged file must only fail its own read)."""
    out = str(tmp_path / "inflate_host_asan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-DINF_HOST", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                            "-I", os.path.join(REPO, "svim_amd", "csrc"), os.path.join(REPO, "tools", "inflate_host_test.cpp"), "-lz", "-o", out],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("no sanitizer runtime in this toolchain")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([out, "--damaged", "120"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "720 streams, 0 wrote behind their output" in run.stdout
    run = subprocess.run([out, "--fuzz", "60"], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0 and "60 buffers, 0 mismatches" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
