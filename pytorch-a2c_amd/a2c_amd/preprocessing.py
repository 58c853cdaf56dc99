"""Host-side per-frame preprocessing with the reference's names (a2c/preprocessing.py).  These
run on the CPU next to the env: pure slicing / thresholding of
one raw frame, no device work.  ``snake_prep`` is out of scope (gym-snake is not in any
BASELINE config).

``breakout_prep`` (preprocessing.py:19-23) calls ``skimage.color.rgb2grey`` on ``pic[::2, ::2, 0]``, which is already a
2-D uint8 array.  scikit-image is unpinned in the reference (requirements.txt:7) and absent from this image, so the
behaviour mirrored here is stated, not measured: ``rgb2grey`` exists in scikit-image <= 0.18 (alias of ``rgb2gray``,
removed in 0.19), and in every one of those releases ``rgb2gray`` starts with ``if rgb.ndim == 2: return
np.ascontiguousarray(rgb)`` -- a 2-D input is returned UNCHANGED (uint8, values 0..255, no division by 255, no
float conversion).  With scikit-image >= 0.19 the reference's import fails outright.  ``breakout_prep`` is therefore the
crop + stride-2 slice of channel 0 as uint8, and its frames travel as uint8 (0..255, not binary: the packed one-bit
transport does not apply).  Both ``pong_prep`` and ``breakout_prep`` are pinned by golden vectors recorded from the REFERENCE's own
functions (tests/golden/g9_*.npz, made by tests/golden/make_golden.py with ``skimage.color`` stubbed by that rule:
tests/test_preprocessing.py on the host functions, tests/test_gpu_frames.py on the device kernel a2c_frame_prep_u8)."""
import numpy as np


def normalize_prep(pic):                       # preprocessing.py:4-6
    return (3 * (pic - 255 / 2) / (255 / 2))[None]


def null_prep(pic):                            # preprocessing.py:8-9
    return pic[None]


def pong_prep(pic):                            # preprocessing.py:11-17
    pic = pic[35:195]
    pic = pic[::2, ::2, 0].copy()
    pic[pic == 144] = 0
    pic[pic == 109] = 0
    pic[pic != 0] = 1
    return pic[None]


def breakout_prep(pic):                        # preprocessing.py:19-23
    pic = pic[35:195, 8:-8]
    pic = pic[::2, ::2, 0]
    return np.ascontiguousarray(pic)[None]
