"""NumPy restatement of the detector post-processing.  TEST INFRASTRUCTURE ONLY.

The reference delegates detection to `ultralytics` (unpinned, absent; weights missing -- SURVEY.md fact 2), so there is
nothing of the reference to pin against: PARITY UNPINNED.  This restates the published YOLOv5 head decode and greedy NMS
that the HIP kernels implement, as the checker for them; the network itself is checked against its own fp32 forward."""
import numpy as np


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float32)))


def decode(raw, anchors, strides, nc, conf_thres):
    """raw: list of float32 [B, 3*(5+nc), H, W] -> boxes [B, T, 4] xyxy, scores [B, T] (-1 when filtered), cls [B, T]."""
    boxes, scores, cls = [], [], []
    for r, a, s in zip(raw, np.asarray(anchors, dtype=np.float32).reshape(3, 3, 2), strides):
        B, _, H, W = r.shape
        y = sigmoid(r.reshape(B, 3, 5 + nc, H, W))
        gy, gx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        cx = (y[:, :, 0] * 2 - 0.5 + gx) * s
        cy = (y[:, :, 1] * 2 - 0.5 + gy) * s
        w = (y[:, :, 2] * 2) ** 2 * a[None, :, 0, None, None]
        h = (y[:, :, 3] * 2) ** 2 * a[None, :, 1, None, None]
        obj, best, bc = y[:, :, 4], y[:, :, 5:].max(axis=2), y[:, :, 5:].argmax(axis=2)
        sc = obj * best
        sc = np.where((obj > conf_thres) & (sc > conf_thres), sc, -1.0)
        boxes.append(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], axis=-1).reshape(B, -1, 4))
        scores.append(sc.reshape(B, -1)); cls.append(bc.reshape(B, -1))
    return np.concatenate(boxes, 1).astype(np.float32), np.concatenate(scores, 1).astype(np.float32), np.concatenate(cls, 1)


def iou(a, b):
    iw = min(a[2], b[2]) - max(a[0], b[0]); ih = min(a[3], b[3]) - max(a[1], b[1])
    inter = max(iw, 0.0) * max(ih, 0.0)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-7)


def nms(boxes, scores, iou_thres, max_det):
    """Greedy NMS over one image's candidates already sorted by descending score -> kept indices."""
    keep = []
    for i in range(len(scores)):
        if scores[i] <= 0 or len(keep) >= max_det:
            break
        if all(iou(boxes[i], boxes[j]) <= iou_thres for j in keep):
            keep.append(i)
    return keep
