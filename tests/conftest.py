import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-rs_amd"))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import q3_oracle
    q3_oracle.build()
    q3_oracle.lib()
    # the test models are small: with one OpenMP thread per core of a 128-core GPU host a forward takes seconds
    q3_oracle.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    return q3_oracle


@pytest.fixture(scope="session")
def np_oracle():
    from oracle import np_oracle as m
    return m


@pytest.fixture(scope="session")
def q3():
    """The product package; loading the HIP library must work even on a CPU-only box (no compute calls)."""
    import qwen3_rs_amd
    import shutil
    import subprocess
    lib = qwen3_rs_amd.lib_path()
    # the build container (where sources are edited) always runs make: a no-op when the library is newer than every
    # source/header the Makefile lists.  Elsewhere (the GPU box gets the prebuilt .so with the snapshot, whose file
    # times are not to be trusted) only a missing library is built.
    dev_box = os.path.isdir("/root/reference")
    if "Q3_HIP_LIB" not in os.environ and shutil.which("make") and (dev_box or not os.path.exists(lib)):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qwen3-rs_amd")])
    qwen3_rs_amd.load_library()
    return qwen3_rs_amd


@pytest.fixture(scope="session")
def tmp_ckpt_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("q3ckpt"))


def golden_path(name):
    return os.path.join(GOLDEN, name)


def bits(a):
    import numpy as np
    return np.ascontiguousarray(a, dtype=np.float32).view(np.int32)


def assert_biteq(a, b, what=""):
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if not np.array_equal(a.view(np.int32), b.view(np.int32)):
        bad = np.nonzero(a.view(np.int32) != b.view(np.int32))[0]
        raise AssertionError(f"{what}: {bad.size}/{a.size} elements differ bitwise, first at {bad[0]}: "
                             f"{a.reshape(-1)[bad[0]]!r} vs {b.reshape(-1)[bad[0]]!r}")
