import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on 8 CPU cores")


def _gpu_available():
    # counting devices does not initialise the GPU runtime in this image
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def anchors():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_anchors.json")))


_oracle_cache = {}


@pytest.fixture(scope="session")
def oracle_scene():
    import oracle_lib as O

    def get(preset):
        if preset not in _oracle_cache:
            _oracle_cache[preset] = O.Scene(preset)
        return _oracle_cache[preset]
    return get


_host_cache = {}


@pytest.fixture(scope="session")
def host_scene():
    import ray_tracer_amd as rt

    def get(preset):
        if preset not in _host_cache:
            _host_cache[preset] = rt.HostScene(preset)
        return _host_cache[preset]
    return get


_ctx_cache = {}


@pytest.fixture(scope="session")
def gpu_ctx(host_scene):
    """A GPU context per preset with the scene uploaded and the LBVH built (shared by the GPU tests)."""
    import ray_tracer_amd as rt

    def get(preset):
        if preset not in _ctx_cache:
            ctx = rt.Context(0)
            ctx.upload(host_scene(preset))
            _ctx_cache[preset] = ctx
        return _ctx_cache[preset]
    yield get
    for c in _ctx_cache.values():
        c.close()
    _ctx_cache.clear()
