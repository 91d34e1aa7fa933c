"""The oracle still gives the answers frozen in tests/golden/oracle_*.npz (made by tests/golden/make_oracle_fixtures.py)."""
import importlib.util
import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")
spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(G, "make_oracle_fixtures.py"))
make = importlib.util.module_from_spec(spec)
spec.loader.exec_module(make)


def test_oracle_reproduces_hit_fixture():
    want = np.load(os.path.join(G, "oracle_hits.npz"))
    got = make.hits_fixture()
    for k in want.files:
        assert np.array_equal(want[k], got[k]), k
    assert (want["node"] >= 0).sum() > 50


def test_oracle_reproduces_image_fixtures():
    want = np.load(os.path.join(G, "oracle_images.npz"))
    got = make.images_fixture()
    for k in want.files:
        assert np.array_equal(want[k], got[k]), k
