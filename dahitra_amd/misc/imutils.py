"""save_image of the reference's misc/imutils.py as models/basic_model.py uses it (a uint8 mask -> PNG)."""
import numpy as np
from PIL import Image


def save_image(image_numpy, image_path):
    arr = np.asarray(image_numpy)
    if arr.dtype != np.uint8:
        arr = np.clip(arr, 0, 255).astype(np.uint8)
    Image.fromarray(arr).save(image_path)
