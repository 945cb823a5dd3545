import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MAT = os.path.join(GOLD, "matrices")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """CPU oracle bindings (test infrastructure)."""
    o = ge.load_oracle()
    o.build()
    return o


@pytest.fixture(scope="session")
def pkg():
    """The product package; the library must already be built (driver runs build() first)."""
    p = ge.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    p.lib()
    return p


@pytest.fixture(scope="session")
def harness(pkg):
    return importlib.import_module(ge.PKG_NAME + ".harness")


@pytest.fixture(scope="session")
def refapi(pkg):
    return importlib.import_module(ge.PKG_NAME + ".refapi")


@pytest.fixture(scope="session")
def kat():
    return json.load(open(os.path.join(GOLD, "kat_traces.json")))


def gold(name):
    return np.load(os.path.join(GOLD, name))


def es_case_fields(row):
    q, nm, nop = int(row[0]), int(row[1]), int(row[2])
    f = lambda i: row[3 + 32 * i: 3 + 32 * i + nm]
    return (q, nm, nop, f(0).view(np.float32).copy(), f(1).view(np.float32).copy(), f(2).copy(), f(3).copy(),
            f(4).view(np.float32).copy(), f(5).copy())


class FieldOracle:
    """An oracle bound to a field order only (for ES/CN cases that need no matrix)."""

    def __init__(self, orc, q):
        self.o = orc.Oracle(os.path.join(MAT, "N96_K48_GF64"))
        assert self.o.L.orc_gf_init(orc.C.byref(self.o.gf), q) == 0
        self.o.q = q
        self.o.idx2bin = np.frombuffer(bytes(self.o.gf.idx2bin), dtype=np.uint8)[:q].copy()
        self.o.bin2idx = np.frombuffer(bytes(self.o.gf.bin2idx), dtype=np.uint8)[:q].copy()

    def __getattr__(self, k):
        return getattr(self.o, k)
