"""The eval-time resize restatement (oracle/resize_oracle.py) is pinned to Pillow itself, bit for bit; the vectorised
coefficient tables of the product (mdqe_cvpr2023_amd/preprocess.py) equal the scalar restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import resize_oracle as RO  # noqa: E402


@pytest.mark.parametrize("shape", [(72, 128, 36, 64), (75, 133, 36, 64), (50, 80, 100, 160), (97, 61, 33, 20), (40, 40, 40, 57),
                                   (123, 77, 123, 30), (90, 160, 45, 80), (37, 53, 90, 131)])
def test_oracle_equals_pillow(shape):
    Image = pytest.importorskip("PIL.Image")
    h, w, oh, ow = shape
    img = np.random.RandomState(h * 1000 + w).randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(RO.resize_bilinear_u8(img, oh, ow), ref)


def test_shortest_edge_rule():
    assert RO.shortest_edge_size(720, 1280, 360, 1333) == (360, 640)
    assert RO.shortest_edge_size(480, 853, 360, 1333) == (360, 640)
    assert RO.shortest_edge_size(1280, 720, 360, 1333) == (640, 360)
    assert RO.shortest_edge_size(300, 3000, 360, 1333) == (133, 1333)          # the long edge caps the scale
    from mdqe_cvpr2023_amd import preprocess as P
    for a in ((720, 1280, 360, 1333), (300, 3000, 360, 1333), (1080, 1920, 480, 1333), (853, 480, 640, 1000)):
        assert P.shortest_edge_size(*a) == RO.shortest_edge_size(*a)


def test_product_tables_equal_the_restatement():
    from mdqe_cvpr2023_amd import preprocess as P
    for a, b in ((128, 64), (133, 64), (80, 160), (61, 20), (40, 57), (1280, 640), (853, 640), (720, 360), (37, 131), (5, 5), (1920, 853)):
        x0, n, k = RO.coeffs(a, b)
        y0, m, q = P.pil_bilinear_coeffs(a, b)
        assert np.array_equal(x0, y0) and np.array_equal(n, m) and np.array_equal(k, q), (a, b)
