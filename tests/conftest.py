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
# The CPU legs of tests/test_zz_full_step_gpu.py cost minutes (a whole 30-block 5B step at L = 9460 in fp32 is 118.8 TFLOP). They are
# started as subprocesses (oracle/step_job.py, 32 host threads each) when a GPU session begins and collected by the tests that sort
# last, so they overlap with the rest of the GPU suite instead of adding their length to it.
_STEP_JOBS = {}


def step_job_result(name, which):
    """block until the oracle forward (case, which) started at session begin is done -> its saved dict."""
    from oracle import step_job
    key = (name, which)
    if key not in _STEP_JOBS:                     # e.g. the test was selected alone with -k after collection: start it now
        _start_step_job(name, which)
    proc, out = _STEP_JOBS[key]
    return step_job.finish_job(proc, out)


def _start_step_job(name, which):
    import tempfile
    from oracle import step_job
    out = os.path.join(tempfile.gettempdir(), f"yume_step_{name}_{which}_{os.getpid()}.pt")
    # (64 threads for the 118.8 TFLOP 5B step, 32 for each of the two 14B forwards: 128 of the GPU box's 256 hardware threads)
    _STEP_JOBS[(name, which)] = (step_job.start_job(name, which, out, threads=64 if name == "5b" else 32), out)


def pytest_collection_finish(session):
    import torch
    if not torch.cuda.is_available():
        return
    wanted = {it.fspath.basename for it in session.items}
    if "test_zz_full_step_gpu.py" in wanted and not session.config.option.collectonly:
        for name, which in (("5b", "cond"), ("14b", "cond"), ("14b", "uncond")):
            _start_step_job(name, which)


def pytest_sessionfinish(session, exitstatus):
    for proc, out in _STEP_JOBS.values():
        if proc.poll() is None:
            proc.kill()
        for p in (out, out + ".log", out + ".tmp"):
            if os.path.exists(p):
                os.remove(p)
