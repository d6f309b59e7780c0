"""Fused acoustic heat-map -> overlay -> detection on one MI355X (BASELINE.json config 4).

Per batch of B (audio window, camera frame) pairs, everything stays in HBM:
  1. delay-and-sum power maps of the B windows                      bf_das_device           (csrc/das_kernels.hip)
  2. colourise, upscale to the camera size, temporal blend, overlay  bf_heatmap_*_device     (csrc/heatmap_kernels.hip)
  3. YOLOv5s-shaped detector on the overlaid frames                  PyTorch-ROCm module graph, convolutions = csrc/conv_kernels.hip
                                                                     (float32 like the reference's predict call; half=True: float16)
  4. head decode + NMS                                               bf_yolo_decode_device / bf_nms_device (csrc/nms_kernels.hip)
In the reference these are three processes joined by queues (beamformer producer, camera reader, YOLO worker; PC/src/
main.pyx:669-736) and the heat-map is blended onto the frame only for display (visual.py:450-455, sensorfusion/
decider.py:44-49); running the detector on the overlaid frame is this build's composition of the same pieces."""
import numpy as np

from interface import config
from lib import _native as nat
import visual
from image_detection.src.yolo_smooth_tracking import Detector


class FusedPipeline:
    def __init__(self, algo="lerp", size=640, device="cuda", half=False):
        import torch
        self.torch, self.device, self.size = torch, device, size
        self.algo = {"pad": nat.PAD, "lerp": nat.LERP}[algo]
        self.mics = None
        self.stream_state = visual.HeatmapStream(size, size, device)
        self.detector = Detector(device=device, half=half)

    def load_tables(self, delays, mics):
        """Steering tables once, as the reference's producer loops do before their frame loop (main.pyx:172-181)."""
        self.mics = np.ascontiguousarray(mics, dtype=np.int32)
        if self.algo == nat.PAD:
            t = np.ascontiguousarray(delays.astype(int).astype(np.int32)).ravel()
            nat.lib.load_coefficients_pad(nat.iptr(t), t.size)
        else:
            t = np.ascontiguousarray(np.float32(delays)).ravel()
            nat.lib.load_coefficients_lerp(nat.fptr(t), t.size)
        nat.check()

    def step(self, d_windows, d_camera, conf_thres=0.1):
        """d_windows float32 [B, N_MICROPHONES, N_SAMPLES]; d_camera uint8 [B, size, size, 3] (BGR).
        Returns (power maps [B, D], overlaid frames uint8 [B, size, size, 3], boxes [B, 300, 6], counts [B])."""
        t = self.torch
        B, D = d_windows.shape[0], config.MAX_RES_X * config.MAX_RES_Y
        power = t.empty((B, D), dtype=t.float32, device=self.device)
        s = t.cuda.current_stream().cuda_stream
        if nat.lib.bf_das_device(self.algo, d_windows.data_ptr(), d_windows.shape[1], power.data_ptr(), D, B, nat.iptr(self.mics), self.mics.size,
                                 0, D, s) != 0:
            nat.check()
        small, _ = self.stream_state.small_heatmaps(power)
        frames = self.stream_state.overlay(small, d_camera)
        boxes, counts = self.detector.detect(frames, conf_thres)
        return power, frames, boxes, counts
