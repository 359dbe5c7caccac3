import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    os.environ.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")      # a GPU fault should fail the test, not write a multi-GB core dump first
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass
    # The torch wheel bundles its own HIP / HSA runtime and libsvx.so links the system one: whichever is initialised FIRST in a process is the
    # one that sees the GPU.  Tests that use torch device tensors beside libsvx need torch to be first (bench.py imports torch first, too).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as om
    om.build()
    return om.Oracle()


# ---- order of the GPU suite --------------------------------------------------------------------------------------------------------
# The driver runs `pytest -m gpu -x`: whatever comes first is what a fault further down cannot hide.  Hot-path parity and the BASELINE configs run first,
# the rank-exchange protocol next, everything that reads BAM files (host reader, GPU inflate, device-resident reader, the life-cycle stress) last.
_READER_WORDS = ("bam", "device_batches", "seek", "bgzf", "inflate", "foreign", "reader", "queryname", "long_cigar_cg", "c3_whole_genome", "bench_harness", "bench_under_torchrun",
                 "bench_two_ranks", "stress")
_RANK_WORDS = ("rank_exchange", "two_ranks", "multigpu_step", "bench_strong")


def gpu_tier(nodeid):
    """0 hot-path parity, 1 BASELINE.json configs at scale, 2 rank exchange, 3 BAM front-ends"""
    name = nodeid.split("::")[-1].lower()
    if any(w in name for w in _READER_WORDS):
        return 3
    if any(w in name for w in _RANK_WORDS):
        return 2
    if "test_gpu_workloads" in nodeid:
        return 1
    return 0


def _workload_rank(nodeid):
    name = nodeid.split("::")[-1]
    for k, w in enumerate(("c1_full_bench_size", "c2_hifi", "c4_clr", "c1_ont_quarter")):      # configs[1] at its full size first
        if w in name:
            return k
    return 9


def pytest_collection_modifyitems(config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if config.pluginmanager.hasplugin("timeout"):
        # a GPU test that hangs (the one A/B run of DESIGN section 10 did: a process group that never came up) must end the run with its name and the stacks of
        # every thread, not sit there until the box's own limit kills a silent process: 15 minutes per test (the slowest takes 35 s), enforced from a watchdog
        # thread because a hang inside a HIP call never returns to the interpreter
        for it in gpu:
            if it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(900, method="thread"))
    if not gpu or os.environ.get("SVX_TEST_ORDER") == "collection":      # (A/B runs of the DESIGN section 10 investigation: the order of round 4)
        return
    order = {id(it): k for k, it in enumerate(items)}
    gpu_sorted = sorted(gpu, key=lambda it: (gpu_tier(it.nodeid), _workload_rank(it.nodeid) if gpu_tier(it.nodeid) == 1 else 0, order[id(it)]))
    slots = [k for k, it in enumerate(items) if it.get_closest_marker("gpu")]
    for k, it in zip(slots, gpu_sorted):
        items[k] = it


@pytest.fixture(autouse=True)
def _device_clean_after_gpu_test(request):
    """every GPU test ends with a check that no kernel or copy of the process has faulted: a fault is sticky and would otherwise be reported by the NEXT
    test's first call - the culprit is named, not the next victim"""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from svim_amd._lib import lib
    L = lib()
    if L.svx_device_synchronize() != 0:
        pytest.fail("the device is not clean after %s: %s" % (request.node.nodeid, L.svx_last_error().decode()), pytrace=False)
