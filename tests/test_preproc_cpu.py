"""Input-transform oracle (oracle/preproc_ref.py).  cv2 is absent from the image, so the
restated OpenCV 8-bit bilinear path is pinned by known answers and against float bilinear
interpolation (PARITY UNPINNED against cv2 itself, see the oracle header)."""
import numpy as np
import torch

from oracle import preproc_ref as pr


def _float_bilinear(img, size):
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    return torch.nn.functional.interpolate(t, size=(size, size), mode='bilinear',
                                           align_corners=False)[0].permute(1, 2, 0).numpy()


def test_identity_and_constant():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (300, 300, 3)).astype(np.uint8)
    assert np.array_equal(pr.resize_linear_u8(img, 300), img)
    flat = np.full((37, 91, 3), 173, np.uint8)
    assert np.all(pr.resize_linear_u8(flat, 512) == 173)


def test_within_one_grey_level_of_float_bilinear():
    rng = np.random.RandomState(1)
    for (h, w, s) in ((375, 500, 300), (500, 333, 512), (120, 77, 300), (1, 1, 8), (2, 3, 7)):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        got = pr.resize_linear_u8(img, s).astype(np.float32)
        assert np.abs(got - _float_bilinear(img, s)).max() <= 1.0, (h, w, s)


def test_known_answer_upsample_2x():
    # 1x2 -> 4 columns: centres at -0.25, 0.25, 0.75, 1.25 -> weights (1,0) (.75,.25) (.25,.75) (0,1)
    img = np.array([[[0, 100, 200], [100, 200, 40]]], np.uint8)
    out = pr.resize_linear_u8(np.repeat(img, 2, 0), 4)
    assert out[0, :, 0].tolist() == [0, 25, 75, 100]
    assert out[3, :, 2].tolist() == [200, 160, 80, 40]


def test_base_transform_layout():
    rng = np.random.RandomState(2)
    img = rng.randint(0, 256, (50, 60, 3)).astype(np.uint8)
    out = pr.base_transform(img, 30, (104, 117, 123))
    assert out.shape == (3, 30, 30) and out.dtype == np.float32
    rs = pr.resize_linear_u8(img, 30)
    assert np.array_equal(out[1], rs[:, :, 1].astype(np.float32) - 117)
