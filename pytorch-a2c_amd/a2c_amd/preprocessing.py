"""Host-side per-frame preprocessing with the reference's names (a2c/preprocessing.py).  These
run on the CPU next to the env: pure slicing / thresholding of
one raw frame, no device work.  ``snake_prep`` is out of scope (gym-snake is not in any
BASELINE config); ``breakout_prep``'s ``rgb2grey`` acts on an already single-channel slice, for
which skimage's rgb2grey is the identity (2-D input), so it is restated as such."""
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
    return np.asarray(pic)[None]
