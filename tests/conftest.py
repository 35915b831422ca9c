import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never returns, a wait that is never answered) must fail as THAT test after ten
    minutes -- the slowest one takes seven seconds -- instead of holding the box until the harness gives up on the whole
    session. pytest-timeout's thread method ends the process, which releases the device."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _hip_library_built():
    """The C-ABI library is a build artefact (git-ignored). Build it on first use so that a fresh checkout can run
    the suite; this is the same call as __graft_entry__.build() and needs only hipcc (no GPU)."""
    from littlemcmc_amd import _build

    if _build.needs_build():
        try:
            _build.build()
        except Exception as err:   # no hipcc on this box: the tests that need the library will say so themselves
            print("could not build liblmc_hip.so: %s" % err)
    yield


@pytest.fixture(scope="session", autouse=True)
def _cold_user_target_cache():
    """On a GPU box every run-time compiled density is compiled THERE: the on-disk cache of hiprtc code objects
    (littlemcmc_amd/_user_targets, git-ignored and .gpurunignore'd) is emptied before the first test, so a -m gpu session
    shows hiprtc working on the MI355X box instead of loading objects the build container happened to leave behind."""
    try:
        import torch

        on_gpu = torch.cuda.is_available()
    except Exception:
        on_gpu = False
    if on_gpu:
        import glob

        for path in glob.glob(os.path.join(ROOT, "littlemcmc_amd", "_user_targets", "*.hsaco")):
            try:
                os.remove(path)
            except OSError:
                pass
    yield
