import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible():
    try:
        import dabtools_amd
        return dabtools_amd.lib().dabhip_device_count() > 0
    except Exception:
        return False      # library not built: the CPU tests that need it fail loudly on their own


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them.  On a GPU box nothing is
    skipped, and when `-m gpu` was asked for explicitly a missing device is an error, not a silent pass."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items or _gpu_visible():
        return
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in config.getoption("-m"):
        raise pytest.UsageError("-m gpu was requested but libdabhip sees no HIP device (there is no CPU fallback to test)")
    skip = pytest.mark.skip(reason="no HIP device visible: GPU parity tests need an MI355X")
    for it in gpu_items:
        it.add_marker(skip)


def record_seed(test_name, seed):
    """The seed of a test that draws a fresh one every run: printed unconditionally (-s or not: written to the real stdout) and appended to
    gpurun_out/test_seeds.txt (merged back from the GPU box), so that a red run is reproducible from more than its assertion text.  Re-run with
    DABHIP_TEST_SEED=<seed> (see fresh_seed)."""
    line = "seed %s %d" % (test_name, seed)
    try:
        sys.__stdout__.write(line + "\n")
        sys.__stdout__.flush()
    except Exception:
        pass
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_seeds.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return seed


def fresh_seed(test_name):
    """A seed from os.urandom -- or DABHIP_TEST_SEED when set, to replay a run -- recorded by record_seed."""
    env = os.environ.get("DABHIP_TEST_SEED")
    seed = int(env) if env and env.isdigit() else int.from_bytes(os.urandom(4), "little")
    return record_seed(test_name, seed)
