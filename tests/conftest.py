import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle as orc  # oracle/oracle.py — builds librt_oracle.so on first use if absent
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def default_scene():
    import raytracing_amd as rt
    return rt.default_scene()


@pytest.fixture(scope="session")
def water_scene():
    import raytracing_amd as rt
    return rt.default_scene(with_water=True)


@pytest.fixture(scope="session")
def gpu_ctx():
    """The session's context.  RTX950_TEST_HEAVY_CLASS=k (read HERE, by the test harness: the library reads no
    environment) soaks the whole suite with every tile of cost class >= k on three-wave workgroups (0 = all)."""
    import raytracing_amd as rt
    if os.environ.get("RTX950_TEST_HEAVY_CLASS"):
        rt.Context.DEFAULT_TUNING[rt.RT_TUNE_HEAVY_CLASS] = int(os.environ["RTX950_TEST_HEAVY_CLASS"])
    ctx = rt.Context(0)
    ctx.reset_tuning()
    yield ctx
    assert ctx.debug_status() == 0, "a heavy tile's workgroup gave up waiting for its partners"
    ctx.close()
