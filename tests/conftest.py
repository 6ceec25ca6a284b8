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
    # A prebuilt library ships with every gpurun snapshot (untracked *.so travel).  It is used only if it was built from
    # the checked-out sources: q3_build_id() (a hash of csrc/* + the header baked in by make) must equal the hash of the
    # sources here; otherwise it is rebuilt, and if that is impossible the session fails instead of testing a stale binary.
    if "Q3_HIP_LIB" not in os.environ:
        want = qwen3_rs_amd.source_build_id()

        # (the id is read in a child process: a stale library must not stay mapped in this one)
        code = ("import ctypes,sys\ntry:\n L=ctypes.CDLL(sys.argv[1]); L.q3_build_id.restype=ctypes.c_char_p; print(L.q3_build_id().decode())\n"
                "except Exception as e: print('unreadable')")
        def read_id():
            if not os.path.exists(lib):
                return None
            return subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True).stdout.strip()
        have = read_id()
        if have != want:
            if not shutil.which("make"):
                raise RuntimeError(f"{lib} was built from other sources (build id {have}, sources {want}) and there is no make to rebuild it")
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qwen3-rs_amd")])
            have = read_id()
            if have != want:
                raise RuntimeError(f"{lib}: build id {have} after make, sources hash to {want}")
    qwen3_rs_amd.load_library()
    return qwen3_rs_amd


_PRODUCT_ENV = {"Q3_PREFILL_M", "Q3_DEBUG_TIMING"}       # what libqwen3_hip.so reads (include/qwen3_hip.h, "Environment")


@pytest.fixture
def dev_forms(q3, monkeypatch):
    """Kernel-form switches live in the developer build only (libqwen3_hip_dev.so, -DQ3_DEV).  `dev_forms({"Q3_X": "1"})` sets the
    variables and, if any of them is not one the product library reads, makes the rest of the test create its engines from the
    developer build (same sources, same results -- which is what these tests assert for every form)."""
    import subprocess
    stack = []

    def apply(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        if any(k not in _PRODUCT_ENV for k in env) and not stack:
            path = q3.dev_lib_path()
            if "Q3_HIP_LIB" not in os.environ:
                # the prebuilt developer library travels with the snapshot like the product library: it is used as it is when its
                # build id matches the sources (no object directory exists on a fresh box, `make` would recompile everything)
                code = ("import ctypes,sys\ntry:\n L=ctypes.CDLL(sys.argv[1]); L.q3_build_id.restype=ctypes.c_char_p; print(L.q3_build_id().decode())\n"
                        "except Exception as e: print('unreadable')")
                have = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True).stdout.strip() if os.path.exists(path) else None
                if have != q3.source_build_id():
                    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "qwen3-rs_amd"), "dev"])
            ctx = q3.use_library(path)
            lib = ctx.__enter__()
            stack.append(ctx)
            have = lib.q3_build_id().decode()
            if "Q3_HIP_LIB" not in os.environ and have != q3.source_build_id():
                raise RuntimeError(f"{path}: build id {have}, sources hash to {q3.source_build_id()}")

    yield apply
    while stack:
        stack.pop().__exit__(None, None, None)


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
