"""Life cycle of the BAM front-ends against the rest of the process: readers are opened, read and closed many times on files of
different sizes - the device-resident reader, also closed in the middle of a pass, and the host reader with GPU inflate - interleaved with device -> host copies of
more than a megabyte into FRESH numpy arrays (svx_memcpy_d2h) and torch `.cpu()` calls.  Every copy is checked byte for byte, every reader's records against
the host reader's, and the device must be clean after every close().  tools/reader_fault_stress.py is the same loop as a command-line tool."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_reader_life_cycle_stress_with_pageable_copies(tmp_path, seed):
    import reader_fault_stress as R
    s = R.stress(iters=200, seed=seed, use_torch=True, tmp_dir=str(tmp_path))
    assert s["iters"] == 200 and s["readers_device"] > 100 and s["copies"] > 300 and s["torch_cpu"] > 100, s
