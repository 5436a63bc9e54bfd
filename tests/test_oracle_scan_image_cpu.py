"""oracle.flat_index.scan_image: the restatement of the image the device's streaming scan reads (float16, or that float16
rounded to its top 12 bits) — known answers for the rounding rule.  The device side of the same rule is pinned against this
function bit for bit in tests/test_index_gpu.py::test_the_scan_images_and_their_error_norms_equal_the_oracles."""
import numpy as np

from oracle import flat_index as orc


def test_twelve_bit_rounding_known_answers():
    x = np.array([[1.0, 1.0 + 2.0 ** -7, 1.0 + 3 * 2.0 ** -8, 1.0 + 2.0 ** -6, -(1.0 + 2.0 ** -7), 1.0 + 3 * 2.0 ** -7,
                   65504.0, -65504.0, 65472.0, 6.1035156e-05, -5.9604645e-08, 0.0, 0.3333333, 1000.5]], dtype=np.float32)
    want = np.array([[1.0, 1.0,                      # a tie goes to the even neighbour ...
                      1.015625, 1.015625, -1.0,
                      1.03125,                       # ... also upwards
                      65024.0, -65024.0, 65024.0,    # what would round up to infinity is truncated
                      6.1035156e-05, -0.0, 0.0,      # a float16 subnormal keeps its top bits; the smallest one rounds to (signed) zero
                      0.33203125, 1000.0]], dtype=np.float32)
    got = orc.scan_image(x, 12)
    assert got.tobytes() == want.tobytes(), (got, want)
    assert orc.scan_image(x, 16).tobytes() == x.astype(np.float16).astype(np.float32).tobytes()


def test_error_norm_of_unit_rows_is_an_eighth_of_a_score_sigma():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2000, 768)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    e12, e16 = orc.scan_image_error(x, 12), orc.scan_image_error(x, 16)
    assert 0.0030 < e12 < 0.0050 and e16 < 0.0004     # ~0.0045 against ~0.0003, sigma = 1 / sqrt(768) = 0.036
    assert orc.scan_image_error(x[:0], 12) == 0.0
