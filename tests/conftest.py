import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The unmodified reference (oracle/_ref) is an OpenMP program that calls LAPACK (the OpenBLAS scipy bundles) from inside its
# parallel loops -- eigen_Mat_rm per keypoint candidate.  That OpenBLAS keeps a fixed table of 128 buffers: on the GPU boxes, with
# a few hundred host threads, the reference overflowed it ("BLAS : Bad memory unallocation!", and once "Program is Terminated.
# Because you tried to allocate too many memory regions" -- after every test had passed).  The checkers run on at most 32 OpenMP
# threads with a single-threaded BLAS under them (set before either library is loaded; bench.py's cpu_baseline is not a test and
# keeps all cores).
if (os.cpu_count() or 1) > 32:                            # (left unset on small hosts: torchrun then gives its workers one thread each)
    os.environ.setdefault("OMP_NUM_THREADS", "32")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The plain-C CPU restatement (test infrastructure, oracle/)."""
    from oracle import oracle as orc
    orc.build(ref=False)
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference compiled under oracle/_ref (skips when absent)."""
    from oracle import oracle as orc
    if not orc.have_ref():
        if os.path.isdir("/root/reference/imutil"):
            orc.build(ref=True)
        if not orc.have_ref():
            pytest.skip("oracle/_ref not built (reference sources not present)")
    return orc.load_ref()


@pytest.fixture(scope="session")
def hip():
    """The product library through the reference's C API (fails loudly when not built / no GPU)."""
    import sift3d_amd
    return sift3d_amd.load()


@pytest.fixture(scope="session")
def hip_testing():
    """The TESTING build of the product library (diagnostic switches and the failure-injection hook compiled in)."""
    import sift3d_amd
    return sift3d_amd.load_testing()
