import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=False)


# ---------------------------------------------------------------------------------------------- full-depth oracle jobs (row N1)
# The CPU legs of tests/test_zz_full_step_gpu.py cost minutes (a whole 30-block 5B step at L = 9460 in fp32 is 118.8 TFLOP). They run as
# subprocesses (oracle/step_job.py), all three side by side, started when THAT module begins (it sorts last) and collected by its tests
# while the device legs run. (Round 4, first attempt: started at session begin to overlap with the whole suite — 128 busy host threads
# next to the other tests' own CPU references, which run on torch's default of every hardware thread, oversubscribed the box: the suite
# went from 5 to 20 minutes.)
_STEP_JOBS = {}


def start_step_jobs():
    for name, which, threads in (("5b", "cond", 96), ("14b", "cond", 32), ("14b", "uncond", 32)):
        if (name, which) not in _STEP_JOBS:
            _start_step_job(name, which, threads)


def step_job_result(name, which):
    """block until the oracle forward (case, which) is done -> its saved dict."""
    from oracle import step_job
    key = (name, which)
    if key not in _STEP_JOBS:
        _start_step_job(name, which, 64)
    proc, out = _STEP_JOBS[key]
    return step_job.finish_job(proc, out)


def _start_step_job(name, which, threads):
    import tempfile
    from oracle import step_job
    out = os.path.join(tempfile.gettempdir(), f"yume_step_{name}_{which}_{os.getpid()}.pt")
    _STEP_JOBS[(name, which)] = (step_job.start_job(name, which, out, threads=threads), out)


def pytest_sessionfinish(session, exitstatus):
    for proc, out in _STEP_JOBS.values():
        if proc.poll() is None:
            proc.kill()
        for p in (out, out + ".log", out + ".tmp"):
            if os.path.exists(p):
                os.remove(p)
