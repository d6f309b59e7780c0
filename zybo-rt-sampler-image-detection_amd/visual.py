"""Device-side counterpart of the reference's heat-map post-processing (PC/src/visual.py), same function names.

`calculate_heatmap(image)` / `find_power_center(image)` take the NumPy power map the reference's functions take and
return what they return, computed by the HIP kernels of csrc/heatmap_kernels.hip through the C-ABI.  `HeatmapStream`
is the batched, device-resident form used by the fused pipeline: power maps in HBM -> colourise -> bilinear upscale ->
0.5/0.5 temporal blend (-> 0.9/0.9 overlay on camera frames) without leaving the GPU (visual.py:440-455)."""
import numpy as np

from interface import config
from lib import _native as nat

WINDOW_DIMENSIONS = (1920, 1080)   # visual.py:9
POWER = 5                           # visual.py:13


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise nat.BeamformerError("no usable HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


class HeatmapStream:
    """Carries the `prev` image of the temporal blend across batches (visual.py:450-451)."""

    def __init__(self, out_w=WINDOW_DIMENSIONS[0], out_h=WINDOW_DIMENSIONS[1], device="cuda"):
        torch = _torch()
        self.out_w, self.out_h, self.device = int(out_w), int(out_h), device
        self.prev = torch.zeros((self.out_h, self.out_w, 3), dtype=torch.uint8, device=device)

    def small_heatmaps(self, d_power, threshold=1e-7, amount=0.5, exponent=POWER):
        """float32 cuda tensor [F, MAX_RES_X*MAX_RES_Y] -> (uint8 [F, MAX_RES_Y, MAX_RES_X, 3], int32 [F] should_overlay)."""
        torch = _torch()
        F = d_power.shape[0]
        small = torch.empty((F, config.MAX_RES_Y, config.MAX_RES_X, 3), dtype=torch.uint8, device=self.device)
        flags = torch.empty((F,), dtype=torch.int32, device=self.device)
        s = torch.cuda.current_stream().cuda_stream
        if nat.lib.bf_heatmap_colorize_device(d_power.data_ptr(), F, threshold, amount, float(exponent), small.data_ptr(), flags.data_ptr(), s) != 0:
            nat.check()
        return small, flags

    def overlay(self, d_small, d_camera=None, w_prev=0.5, w_new=0.5, w_cam=0.9, w_heat=0.9):
        """uint8 [F, MAX_RES_Y, MAX_RES_X, 3] (+ optional camera frames uint8 [F, H, W, 3]) -> uint8 [F, H, W, 3]."""
        torch = _torch()
        F = d_small.shape[0]
        out = torch.empty((F, self.out_h, self.out_w, 3), dtype=torch.uint8, device=self.device)
        s = torch.cuda.current_stream().cuda_stream
        cam = d_camera.data_ptr() if d_camera is not None else None
        if nat.lib.bf_heatmap_overlay_device(d_small.data_ptr(), F, self.out_w, self.out_h, self.prev.data_ptr(), cam, out.data_ptr(),
                                             w_prev, w_new, w_cam, w_heat, s) != 0:
            nat.check()
        return out


def calculate_heatmap(image, threshold=1e-7, amount=0.5, exponent=POWER, window=WINDOW_DIMENSIONS):
    """visual.py:143-188: power map [MAX_RES_X, MAX_RES_Y] -> (heatmap uint8 [H, W, 3], should_overlay)."""
    torch = _torch()
    img = np.ascontiguousarray(image[..., 0] if image.ndim == 3 else image, dtype=np.float32)
    st = HeatmapStream(window[0], window[1])
    small, flags = st.small_heatmaps(torch.from_numpy(img.reshape(1, -1)).cuda(), threshold, amount, exponent)
    out = st.overlay(small, None, w_prev=0.0, w_new=1.0)
    return out[0].cpu().numpy(), bool(flags[0].item())


def find_power_center(image, region_size=3):
    """visual.py:295-322 -> (center_x, center_y)."""
    torch = _torch()
    img = np.ascontiguousarray(image, dtype=np.float32)
    d = torch.from_numpy(img.reshape(1, -1)).cuda()
    centers = torch.empty((1, 2), dtype=torch.float32, device="cuda")
    ws = torch.empty_like(d)
    if nat.lib.bf_power_center_device(d.data_ptr(), 1, centers.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream) != 0:
        nat.check()
    c = centers.cpu().numpy()[0]
    return float(c[0]), float(c[1])
