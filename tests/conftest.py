import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The unmodified reference (oracle/_ref) reaches LAPACK through the OpenBLAS that scipy bundles.  On the many-core GPU boxes that
# library's thread pool reported "BLAS : Bad memory unallocation!" during the registration tests and once took the interpreter
# down at exit, after every test had passed.  The checker needs no BLAS threads (set before the library is first loaded).
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
