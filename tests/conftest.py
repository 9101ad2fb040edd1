import copy
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def desc():
    from hpt import scene_io
    return scene_io.load_scene(os.path.join(GOLDEN, "scene.xml"))


@pytest.fixture(scope="session")
def olib():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def oscene(olib, desc):
    import oracle_lib
    return oracle_lib.OracleScene(olib, desc)


def resized(desc, w, h):
    d = copy.deepcopy(desc)
    d.camera.width, d.camera.height = w, h
    return d


@pytest.fixture(scope="session")
def ctx():
    import hpt
    c = hpt.Context(0)
    yield c


def host_tree(m):
    from hpt import host_bvh
    return host_bvh.build_reference_bvh(m.verts, m.tris)


@pytest.fixture(scope="session")
def gscene(ctx, desc):
    import hpt
    return hpt.Scene(ctx, desc, host_bvh=host_tree)
