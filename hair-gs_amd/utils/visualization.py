"""Image-space helpers of the render CLI (reference utils/visualization.py:251-269)."""
import numpy as np
import torch


def orientation_map_to_vis(orientation_map, confidence_map) -> np.ndarray:
    """[H,W] angles in radians -> uint8 RGB where hue = angle (full saturation and value; OpenCV's 8-bit HSV convention
    H = degrees / 2 in 0..179, as the reference obtains through cv2.cvtColor); pixels with confidence == 1 are black."""
    if isinstance(orientation_map, torch.Tensor):
        orientation_map = orientation_map.squeeze().detach().cpu().numpy()
    if isinstance(confidence_map, torch.Tensor):
        confidence_map = confidence_map.detach().cpu().numpy()
    h8 = (180.0 * orientation_map / np.pi).astype(np.uint8)            # the reference's uint8 cast of the hue channel
    hue = (h8.astype(np.float32) * 2.0) % 360.0 / 60.0                  # sector position in [0, 6)
    x = 1.0 - np.abs(hue % 2.0 - 1.0)
    sector = np.floor(hue).astype(np.int32) % 6
    one, zero = np.ones_like(x), np.zeros_like(x)
    table = [(one, x, zero), (x, one, zero), (zero, one, x), (zero, x, one), (x, zero, one), (one, zero, x)]
    rgb = np.zeros(orientation_map.shape + (3,), np.float32)
    for k, (r, g, b) in enumerate(table):
        m = sector == k
        rgb[m] = np.stack([r[m], g[m], b[m]], axis=-1)
    out = np.rint(rgb * 255.0).astype(np.uint8)
    out[confidence_map == 1.0] = 0
    return out
