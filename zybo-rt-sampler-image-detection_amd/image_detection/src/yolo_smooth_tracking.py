"""Mirror of the detection caller of the reference (image-detection/src/yolo_smooth_tracking.py): `yolo_model(path)
.get_detections(frame, conf_threshold)` -> [[x1, y1, x2, y2, conf], ...] and `compute_iou`, with the network forward in
PyTorch-ROCm -- its 60 convolutions (+ bias + SiLU) in the implicit-GEMM MFMA kernels of csrc/conv_kernels.hip, float32 like the
reference's predict call or float16 -- and the head decode + candidate selection + NMS in the HIP kernels of csrc/nms_kernels.hip.

`model_path` may name a state_dict saved from `image_detection.model.yolov5s.YOLOv5s`; with None (the reference's
weights are not in its repository) a seeded random-init network is used.  SORT tracking (sort/sort.py, GPL, CPU) and the
OpenCV drawing of `process_video_track_boxes_only` stay with the reference; `split_detections` is its banding logic."""
import ctypes as C
import os

import numpy as np

from lib import _native as nat
from ..model import yolov5s

CONF_HIGH, CONF_LOW = 0.5, 0.1          # yolo_smooth_tracking.py:278-279
IOU_THRES, MAX_DET, MAX_NMS = 0.45, 300, 1024


class Detector:
    """Batched device-resident detector: uint8/float frames already on the GPU in, [B, MAX_DET, 6] boxes + counts out.

    half=False (the default) is the reference's precision: `ultralytics.YOLO(...).predict(frame)` runs float32 unless told
    otherwise (yolo_smooth_tracking.py:13-23 passes no half=).  half=True is the fast mode (float16 tensors, f32 accumulation).
    Either way the convolutions are the library's own MFMA kernels (conv_backend "hip"; "miopen" = torch's, kept as an A/B baseline).

    The buffers whose raw addresses go into kernel arguments (decode / candidate / NMS workspaces and the returned boxes / counts) are
    allocated once per (shape, stream) and reused: a captured graph replays onto memory this object owns, and two streams never share
    a workspace.  The tensors returned by postprocess / detect are therefore valid until the next call with the same shape on the same
    stream -- clone them to keep them longer."""

    def __init__(self, model_path=None, nc=1, seed=0, device="cuda", half=False, conv_backend=None):
        import torch
        if not torch.cuda.is_available():
            raise nat.BeamformerError("no usable HIP device; the detector has no CPU path")
        self.torch, self.device, self.half, self.nc = torch, device, half, nc
        backend = conv_backend or os.environ.get("BF_CONV_BACKEND", "hip")
        self.net = yolov5s.build(nc, seed, device, half, "miopen")
        if model_path is not None:
            self.net.load_state_dict(torch.load(model_path, map_location=device))      # (before the weights are repacked for the kernel)
        if backend == "hip":
            self.net = yolov5s.use_hip_convs(self.net)
        elif backend != "miopen":
            raise ValueError("conv_backend: 'miopen' or 'hip'")
        self.conv_backend = backend
        self.dtype = torch.float16 if half else torch.float32
        self.anchors = np.ascontiguousarray(np.asarray(yolov5s.ANCHORS, dtype=np.float32).reshape(3, 3, 2))
        self._ws = {}

    def preprocess(self, frames_u8):
        """uint8 [B, H, W, 3] (BGR as OpenCV delivers it, main.pyx:632) -> network input [B, 3, H, W] RGB in [0, 1], channels_last
        ([B, 4, H, W] with a zero fourth channel on the HIP convolution path)."""
        t = self.torch
        if self.conv_backend == "hip" and frames_u8.dtype == t.uint8 and frames_u8.is_contiguous():
            # one kernel, straight into the 4-channel NHWC buffer the stem convolution reads (channel 3 zero)
            b, h, w, _ = frames_u8.shape
            x = t.empty((b, 4, h, w), dtype=self.dtype, device=frames_u8.device, memory_format=t.channels_last)
            fn = nat.lib.bf_preprocess_bgr8_device if self.half else nat.lib.bf_preprocess_bgr8_f32_device
            if fn(frames_u8.data_ptr(), x.data_ptr(), b, h, w, 4, t.cuda.current_stream().cuda_stream) != 0:
                nat.check()
            return x
        x = frames_u8.flip(-1).permute(0, 3, 1, 2)
        x = (x.half() if self.half else x.float()) / 255
        return x.contiguous(memory_format=t.channels_last)

    def raw(self, x):
        """The three head maps as the network leaves them: [B, 3 * (5 + nc), H, W] tensors in channels_last memory."""
        with self.torch.no_grad():
            return self.net(x)

    def postprocess(self, raw, conf_thres=CONF_LOW, iou_thres=IOU_THRES, max_det=MAX_DET):
        """Head maps -> (boxes float32 [B, max_det, 6] = x1, y1, x2, y2, conf, cls;  counts int32 [B]) on the device."""
        t = self.torch
        B = raw[0].shape[0]
        cl = all(r.is_contiguous(memory_format=t.channels_last) for r in raw)
        if not cl:
            raw = [r.contiguous() for r in raw]
        hs = (C.c_int * 3)(*[int(r.shape[2]) for r in raw]); ws = (C.c_int * 3)(*[int(r.shape[3]) for r in raw])
        st = (C.c_int * 3)(*yolov5s.STRIDES)
        ptrs = (C.c_void_p * 3)(*[r.data_ptr() for r in raw])
        T = 3 * sum(int(r.shape[2]) * int(r.shape[3]) for r in raw)
        K = min(MAX_NMS, T)
        s = t.cuda.current_stream().cuda_stream
        key = (B, T, K, max_det, s)
        ws_ = self._ws.get(key)
        if ws_ is None:
            ws_ = self._ws[key] = dict(boxes=t.empty((B, T, 4), dtype=t.float32, device=self.device),
                                       scores=t.empty((B, T), dtype=t.float32, device=self.device),
                                       cls=t.empty((B, T), dtype=t.int32, device=self.device),
                                       mask=t.empty((B, K, (K + 63) // 64), dtype=t.int64, device=self.device),
                                       top=t.empty((B, K), dtype=t.float32, device=self.device), cb=t.empty((B, K, 4), dtype=t.float32, device=self.device),
                                       cc=t.empty((B, K), dtype=t.int32, device=self.device), counts=t.empty((B,), dtype=t.int32, device=self.device),
                                       out=t.empty((B, max_det, 6), dtype=t.float32, device=self.device), n_out=t.empty((B,), dtype=t.int32, device=self.device))
        boxes, scores, cls, mask = ws_["boxes"], ws_["scores"], ws_["cls"], ws_["mask"]
        fmt = (1 if raw[0].dtype == t.float16 else 0) | (2 if cl else 0)          # bit 0: float16 maps, bit 1: NHWC (channels_last) maps
        if nat.lib.bf_yolo_decode_device(ptrs, hs, ws, st, nat.fptr(self.anchors), B, self.nc, fmt, conf_thres, boxes.data_ptr(),
                                         scores.data_ptr(), cls.data_ptr(), s) != 0:
            nat.check()
        top, cb, cc, counts = ws_["top"], ws_["cb"], ws_["cc"], ws_["counts"]
        if nat.lib.bf_topk_candidates_device(scores.data_ptr(), boxes.data_ptr(), cls.data_ptr(), B, T, K, top.data_ptr(), cb.data_ptr(), cc.data_ptr(),
                                             counts.data_ptr(), s) != 0:        # candidate order for the greedy pass
            nat.check()
        out, n_out = ws_["out"], ws_["n_out"]          # (the scan kernel writes every row: kept boxes, then zeros)
        if nat.lib.bf_nms_device(cb.data_ptr(), top.data_ptr(), cc.data_ptr(), counts.data_ptr(), B, K, iou_thres, max_det, mask.data_ptr(),
                                 out.data_ptr(), n_out.data_ptr(), s) != 0:
            nat.check()
        return out, n_out

    def detect(self, frames_u8, conf_thres=CONF_LOW):
        return self.postprocess(self.raw(self.preprocess(frames_u8)), conf_thres)


def letterbox_geometry(h, w, imgsz=640, stride=32):
    """ultralytics' LetterBox(imgsz, auto=True, scaleup=True, center=True) for an h x w frame -> (new_h, new_w, top, left, out_h, out_w, gain):
    the long side is scaled to imgsz (up or down), the short side padded to the next multiple of the stride, the picture centred."""
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = (imgsz - new_w) % stride, (imgsz - new_h) % stride
    top, left = int(round(dh / 2 - 0.1)), int(round(dw / 2 - 0.1))
    bottom, right = int(round(dh / 2 + 0.1)), int(round(dw / 2 + 0.1))
    return new_h, new_w, top, left, new_h + top + bottom, new_w + left + right, r


class yolo_model:
    """yolo_smooth_tracking.py:9-23"""

    def __init__(self, model_path=None):
        self.model = Detector(model_path)

    def get_detections(self, frame, conf_threshold=0.0):
        """frame: uint8 [H, W, 3] BGR -> [[x1, y1, x2, y2, conf], ...] in the frame's pixel coordinates, conf >= conf_threshold.
        The frame is letterboxed the way ultralytics' predict does it (long side scaled to 640 with cv2.resize(INTER_LINEAR) semantics, the
        short side centred in the next multiple of 32 on a grey (114) border: the reference's 360 x 640 camera frames, main.pyx:632, run
        at scale 1 in a 384 x 640 canvas); the boxes are shifted and scaled back and clipped to the frame (scale_boxes)."""
        t = self.model.torch
        frame = np.ascontiguousarray(frame)
        H, W = int(frame.shape[0]), int(frame.shape[1])
        new_h, new_w, top, left, Hp, Wp, gain = letterbox_geometry(H, W)
        f = t.from_numpy(frame).to(self.model.device)
        if (Hp, Wp) != (H, W):
            canvas = t.empty((Hp, Wp, 3), dtype=t.uint8, device=self.model.device)
            if nat.lib.bf_letterbox_bgr8_device(f.data_ptr(), H, W, canvas.data_ptr(), Hp, Wp, new_h, new_w, top, left, 114, t.cuda.current_stream().cuda_stream) != 0:
                nat.check()
            f = canvas
        out, n = self.model.detect(f.unsqueeze(0), conf_thres=max(1e-3, min(conf_threshold, 0.25)) if conf_threshold > 0 else 0.25)
        out, n = out[0].cpu().numpy(), int(n[0].item())
        dets = []
        for i in range(n):
            if out[i, 4] < conf_threshold:
                continue
            x1, y1, x2, y2 = (float(v) for v in out[i, :4])
            x1, x2 = min(max((x1 - left) / gain, 0.0), float(W)), min(max((x2 - left) / gain, 0.0), float(W))
            y1, y2 = min(max((y1 - top) / gain, 0.0), float(H)), min(max((y2 - top) / gain, 0.0), float(H))
            dets.append([x1, y1, x2, y2, float(out[i, 4])])
        return dets


def compute_iou(box1, box2):
    """yolo_smooth_tracking.py:26-37"""
    x1, y1, x2, y2 = box1
    x1g, y1g, x2g, y2g = box2
    inter = max(0, min(x2, x2g) - max(x1, x1g)) * max(0, min(y2, y2g) - max(y1, y1g))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    return inter / union if union > 0 else 0


def split_detections(detections, confh=CONF_HIGH, confl=CONF_LOW):
    """yolo_smooth_tracking.py:303-305: (valid, candidates) bands handed to the tracker."""
    return [d for d in detections if d[4] > confh], [d for d in detections if confl < d[4] <= confh]
