import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_bindings import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="session")
def g2_path(golden_dir, tmp_path_factory):
    """tests/golden/g2.sfx.xz (the repeat-family genome of the `-r3/-r4` cases) unpacked to a temp file"""
    import lzma

    p = tmp_path_factory.mktemp("golden") / "g2.sfx"
    with lzma.open(os.path.join(golden_dir, "g2.sfx.xz"), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    return str(p)


@pytest.fixture(scope="session")
def g1_el5_path(golden_dir, tmp_path_factory):
    """tests/golden/g1_el5.sfx.xz unpacked to a temp file"""
    import lzma

    p = tmp_path_factory.mktemp("golden") / "g1_el5.sfx"
    with lzma.open(os.path.join(golden_dir, "g1_el5.sfx.xz"), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    return str(p)


def _unxz(golden_dir, tmp_path_factory, name):
    import lzma

    p = tmp_path_factory.mktemp("golden") / name
    with lzma.open(os.path.join(golden_dir, name + ".xz"), "rb") as f, open(p, "wb") as g:
        g.write(f.read())
    return str(p)


@pytest.fixture(scope="session")
def g3_path(golden_dir, tmp_path_factory):
    """tests/golden/g3.sfx.xz (the genome of the optional-phase vectors: planted introns) unpacked to a temp file"""
    return _unxz(golden_dir, tmp_path_factory, "g3.sfx")


@pytest.fixture(scope="session")
def g3_el5_path(golden_dir, tmp_path_factory):
    return _unxz(golden_dir, tmp_path_factory, "g3_el5.sfx")
