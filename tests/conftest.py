import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch's default intra-op pool follows os.cpu_count() (256 on a GPU box whose cgroup grants 16 CPUs): throttled
    # 5x (msf_wsi_amd/hostcpu.py).  Size it to what the process may really use.
    from msf_wsi_amd.hostcpu import set_torch_threads, usable_cpus

    # processes the tests spawn (the rank workers of test_dist_gpu / test_dropin_gpu: 2-4 at a time, and whatever else
    # inherits this environment) start with a quarter of the usable CPUs each instead of os.cpu_count() / 2 threads
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, usable_cpus() // 4)))
    set_torch_threads()  # this process: an explicit torch.set_num_threads overrides OMP_NUM_THREADS


# ----------------------------------------------------------------------------------------------------------------
# Order of the GPU suite: the whole-step parity, trainer, multi-rank and 16-bit tests run FIRST,
# the parametrised kernel sweeps (test_kernels_gpu: ~390 cases, test_production_gpu: ~100) LAST, so that a time limit
# can never again cut off the parity end of the suite.  Inside the first group the tests that wait for the long CPU
# oracle jobs (224x224 config 1, the ResNet-50-derived model, the 30-step bf16 curve) come last: the background workers
# (tests/oracle_jobs.py) compute those while the GPU runs everything before them.
# ----------------------------------------------------------------------------------------------------------------
FILE_ORDER = ["test_parity_gpu", "test_train_gpu", "test_dist_gpu", "test_dropin_gpu", "test_fixes_gpu",
              "test_lowp_parity_gpu", "test_lowp_default_gpu", "test_blocks_lowp_gpu", "test_encoder_gpu", "test_headline_geometry_gpu", "test_options_gpu",
              "test_hooknet_gpu", "test_tiler",
              "test_augment", "test_metrics", "test_kernels_gpu", "test_production_gpu"]
# (substring of the node id, rank inside the first group): tests that need a long-running oracle job
LATE = [("test_step_parity_r18_b8_s224_config1", 1), ("test_step_parity_r50_b8_s64_diverse", 2),
        ("test_lowp_step_within_reference_autocast_spread_r50", 2), ("test_default_path_lowp_r50", 2),
        ("test_loss_curve_tracks_reference", 3)]
FIRST_GROUP_END = FILE_ORDER.index("test_hooknet_gpu")


WIDE = "test_ranks_match_single_process[4]"  # 4 ranks + the test process = 5 of the 6 processes a box lets use its GPU


def _rank(item):
    mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    fi = FILE_ORDER.index(mod) if mod in FILE_ORDER else len(FILE_ORDER)
    if WIDE in item.nodeid:
        return (-1, 0, 0)              # the very first test: before the oracle workers exist (see _start_workers)
    for sub, r in LATE:
        if sub in item.nodeid:
            return (1, r, fi)          # after the first group's other tests, before the sweeps
    if fi <= FIRST_GROUP_END:
        return (0, fi, 0)
    return (2, fi, 0)


def pytest_collection_modifyitems(session, config, items):
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    order = {id(it): i for i, it in enumerate(items)}
    items.sort(key=lambda it: (_rank(it), order[id(it)]) if it.get_closest_marker("gpu") is not None
               else ((-1, 0, 0), order[id(it)]))


# the CPU oracle jobs the GPU suite consumes, longest first (tests/oracle_jobs.py)
# (two workers: the first takes the long ResNet-50-derived case at once, the second works through the short jobs in the
#  order the first tests ask for them and then the 224x224 case and the 30-step curve)
ORACLE_REQUESTS = [
    ("oracle_case", ("r50_b8_s64_div",)),
    ("oracle_case", ("r18_b8_s64",)),
    ("oracle_case", ("r18_b16_s64_div",)),
    ("steps", ("r18_b8_s64", 2, "fp64")),
    ("steps", ("r18_b8_s64", 2, "fp32")),
    ("steps", ("r18_b8_s64", 1, "fp64", "infonce", 0.2, True)),
    ("steps", ("r18_b8_s64", 1, "fp32", "infonce", 0.2, True)),
    ("oracle_case", ("r18_b8_s224",)),
    ("curve_oracle", ("r18_b16_s64_curve", "bf16")),
]


_WANT = None


def pytest_collection_finish(session):
    """decide which background oracle jobs this run needs -- only for a run that selected GPU tests on a machine with a
    GPU (the CPU suite computes what it needs inline).  The workers themselves start in pytest_runtest_setup."""
    global _WANT
    if os.environ.get("MSFWSI_TEST_WORKERS", "1") == "0":
        return
    sel = [it for it in session.items if it.get_closest_marker("gpu") is not None]
    if len(sel) < 20:   # a hand-picked few tests: inline is fine
        return
    try:
        import torch

        if torch.cuda.device_count() == 0:
            return
    except Exception:
        return
    needed = {it.nodeid for it in sel}
    _WANT = []
    for name, key in ORACLE_REQUESTS:
        case = key[0]
        hint = {"r50_b8_s64_div": "r50", "r18_b8_s224": "s224", "r18_b16_s64_curve": "curve"}.get(case)
        if hint is None or any(hint in n for n in needed):
            _WANT.append((name, key))


@pytest.hookimpl(tryfirst=True)
def pytest_runtest_setup(item):
    """Start the oracle workers before the first test that is not the 4-rank rehearsal.  A worker process ends up with
    the GPU device nodes open although it computes on the CPU only and sees no device (torch's autograd engine counts the
    HIP devices at its first backward, which initialises the runtime), so it counts against the box's limit of 6
    processes per GPU: 4 ranks + the test process + 2 workers would be 7.  The 4-rank test is ordered first and runs
    before the workers exist; every other multi-rank test has 2 ranks (2 + 1 + 2 workers = 5)."""
    global _WANT
    if _WANT is None or WIDE in item.nodeid:
        return
    import oracle_jobs

    want, _WANT = _WANT, None
    oracle_jobs.start(want, workers=2)


def pytest_sessionfinish(session, exitstatus):
    if "oracle_jobs" in sys.modules:
        sys.modules["oracle_jobs"].stop()


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load the C-ABI library; never falls back to anything else."""
    from msf_wsi_amd import _lib

    _lib.build()
    return _lib.load()


@pytest.fixture
def reproducible_sums(hip_lib):
    """Whole-model statistical tests (16-bit parity gates, the trunk's three-seed rule, the loss curve) run with ONE
    workgroup per weight-gradient tile (msfwsi_set_tuning(15, 1)): the default's pixel splits add their partial sums with
    fp32 atomics in arrival order, and 16-bit storage amplifies that 4e-7 to a 1e-2 different gradient between two runs of
    the ResNet-50-derived step (tools/race_check.py) -- enough to tip a marginal gate one way on one run and the other way
    on the next.  With the cap every value of the step repeats bit for bit, so these tests have one outcome per build.
    The kernel-level tests (test_kernels_gpu, test_production_gpu, test_train_gpu, ...) keep the default split-K path, and
    test_lowp_parity_gpu.py::test_default_path_* gate the DEFAULT configuration three runs in a row."""
    from helpers import tuned

    with tuned(hip_lib, {15: 1}):
        yield
