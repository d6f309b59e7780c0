"""YOLOv5s-shaped detector (PyTorch-ROCm) -- the network behind the reference's `ultralytics.YOLO(best.pt)`.

The reference ships neither the architecture nor the weights (`image-detection/model/*` is git-ignored and listed in
.MISSING_LARGE_BLOBS; `ultralytics` is an unpinned third-party package), so this is the published YOLOv5s v6.x graph
(depth 0.33, width 0.50: 6x6/2 stem, C3 stages 64-128-256-512, SPPF, PAN head, three detect levels at strides 8/16/32,
anchors of the COCO release) with seeded random weights and `nc` classes (1 in the reference's use: drones).
Convolutions (channels_last, BatchNorm folded; float32 -- the precision ultralytics' predict runs at -- or float16) run through the
library's own implicit-GEMM MFMA kernel (`HipConv`, csrc/conv_kernels.hip: conv + bias + SiLU in one launch; float32 -- on the bfloat16 matrix pipes through an exact three-way operand split, or on the float32 matrix instruction -- or f16
matrix instructions) or, with conv_backend="miopen", through torch / MIOpen; the head decode and the NMS are the hand-written HIP
kernels of csrc/nms_kernels.hip.  On the HIP path no concatenation is ever materialised: convolutions write channel slices of the
buffer their consumer reads, and the 1x1 layers behind the head's cat((upsample(a), b)) / cat((conv(x), b)) read both sources."""
import torch
import torch.nn as nn

ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
STRIDES = [8, 16, 32]


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, p=None):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU(inplace=True)

    def forward(self, x, **into):
        """`into` (HipConv only): out= a channel slice of a wider channels_last buffer to write to, residual= a tensor to add."""
        return self.act(self.bn(self.conv(x, **into))) if into else self.act(self.bn(self.conv(x)))

    def fuse(self):
        """Fold the BatchNorm into the convolution (inference)."""
        w = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
        fused = nn.Conv2d(self.conv.in_channels, self.conv.out_channels, self.conv.kernel_size, self.conv.stride, self.conv.padding, bias=True)
        fused.weight.data = (self.conv.weight * w.view(-1, 1, 1, 1)).detach()
        fused.bias.data = (self.bn.bias - self.bn.running_mean * w).detach()
        self.conv, self.bn = fused, nn.Identity()


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True):
        super().__init__()
        self.cv1, self.cv2 = Conv(c1, c2, 1), Conv(c2, c2, 3)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        if isinstance(self.cv2.conv, HipConv):            # the add (and the caller's concatenation) happen in cv2's epilogue
            return self.cv2(self.cv1(x), out=out, residual=x if self.add else None)
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class C3(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True):
        super().__init__()
        c_ = c2 // 2
        self.cv1, self.cv2, self.cv3 = Conv(c1, c_, 1), Conv(c1, c_, 1), Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut) for _ in range(n)))

    def forward(self, x):
        """x: a tensor, or (a, b, up) = the concatenation torch.cat((upsample2x(a) if up else a, b), 1), which the HIP path never builds."""
        hip = isinstance(self.cv3.conv, HipConv)
        src = {}
        if isinstance(x, tuple):
            a, b2, up = x
            if hip:
                x, src = a, dict(x2=b2, up=up)
            else:
                x = torch.cat((nn.functional.interpolate(a, scale_factor=2, mode="nearest") if up else a, b2), 1)
        if hip:
            # both halves of the block's own concatenation are convolution outputs: they are written straight into its buffer
            ref = src["x2"] if src else x
            b, h, w = int(ref.shape[0]), int(ref.shape[2]), int(ref.shape[3])
            c_ = self.cv1.conv.n
            buf = torch.empty((b, 2 * c_, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            self.cv2(x, out=buf[:, c_:], **src)
            y = self.cv1(x, **src)
            for blk in self.m[:-1]:
                y = blk(y)
            self.m[-1](y, out=buf[:, :c_])
            return self.cv3(buf)
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1, self.cv2 = Conv(c1, c_, 1), Conv(c_ * 4, c2, 1)
        self.m = nn.MaxPool2d(k, 1, k // 2)

    def forward(self, x):
        if isinstance(self.cv1.conv, HipConv) and x.shape[2] * x.shape[3] <= 2048:
            # cv1 writes the first quarter of the concatenation buffer, one kernel fills the other three with the cascaded pools
            from lib import _native as nat
            b, _, h, w = (int(v) for v in x.shape)
            c_ = self.cv1.conv.n
            buf = torch.empty((b, 4 * c_, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            self.cv1(x, out=buf[:, :c_])
            pool = nat.lib.bf_sppf_pool_f32_device if x.dtype == torch.float32 else nat.lib.bf_sppf_pool_device
            if pool(buf.data_ptr(), b, h, w, c_, torch.cuda.current_stream().cuda_stream) != 0:
                nat.check()
            return self.cv2(buf)
        x = self.cv1(x)
        y1 = self.m(x)
        y2 = self.m(y1)
        return self.cv2(torch.cat((x, y1, y2, self.m(y2)), 1))


class YOLOv5s(nn.Module):
    """forward(x [B, 3, H, W] in [0, 1]) -> the three raw head maps [B, 3*(5+nc), H/8, W/8], [.., H/16, ..], [.., H/32, ..]."""

    def __init__(self, nc=1):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.b0, self.b1, self.b2 = Conv(3, 32, 6, 2, 2), Conv(32, 64, 3, 2), C3(64, 64, 1)
        self.b3, self.b4 = Conv(64, 128, 3, 2), C3(128, 128, 2)
        self.b5, self.b6 = Conv(128, 256, 3, 2), C3(256, 256, 3)
        self.b7, self.b8, self.b9 = Conv(256, 512, 3, 2), C3(512, 512, 1), SPPF(512, 512)
        self.h10, self.h13 = Conv(512, 256, 1), C3(512, 256, 1, False)
        self.h14, self.h17 = Conv(256, 128, 1), C3(256, 128, 1, False)
        self.h18, self.h20 = Conv(128, 128, 3, 2), C3(256, 256, 1, False)
        self.h21, self.h23 = Conv(256, 256, 3, 2), C3(512, 512, 1, False)
        self.detect = nn.ModuleList(nn.Conv2d(c, 3 * self.no, 1) for c in (128, 256, 512))

    def forward(self, x):
        p3 = self.b4(self.b3(self.b2(self.b1(self.b0(x)))))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        t10 = self.h10(p5)
        # (a, b, up): torch.cat((self.up(a) if up else a, b), 1) -- C3 reads the two sources itself on the HIP path
        t14 = self.h14(self.h13((t10, p4, True)))
        o3 = self.h17((t14, p3, True))
        o4 = self.h20((self.h18(o3), t14, False))
        o5 = self.h23((self.h21(o4), t10, False))
        return [d(o) for d, o in zip(self.detect, (o3, o4, o5))]

    def fuse(self):
        for m in self.modules():
            if isinstance(m, Conv) and isinstance(m.bn, nn.BatchNorm2d):
                m.fuse()
        return self


def _nhwc_view(t, b, c, h, w, dtype):
    """(pointer, elements between consecutive pixels) of a [b, c, h, w] channels_last tensor or channel slice of one, else None."""
    if t is None or tuple(t.shape) != (b, c, h, w) or t.dtype != dtype or t.stride(1) != 1 or t.stride(2) != w * t.stride(3) or t.stride(0) != h * w * t.stride(3):
        return None
    return t.data_ptr(), int(t.stride(3))


class HipConv(nn.Module):
    """A folded convolution (+ SiLU) through the library's own kernel (csrc/conv_kernels.hip, implicit GEMM on the matrix cores):
    bf_conv2d_nhwc_f32_* for a float32 convolution (float32 operands, exact products, float32 sums; bf_conv2d_f32_mode picks the matrix instruction), bf_conv2d_nhwc_f16_* for a float16 one.
    In and out: [B, C, H, W] tensors of the convolution's dtype in channels_last memory, i.e. NHWC buffers.
    The weights are repacked once to [N][KH][KW][C'] with C' = the input channels padded to what the kernel takes (the 3-channel
    image of the 6x6 stem becomes 4 channels, the rest are powers of two already), each row zero-padded to whole K stages.
    The packed weights and the bias are plain attributes, not module buffers: a later net.half() / .float() must not recast what the
    kernel reads by raw pointer (convert the network first, then call use_hip_convs)."""

    def __init__(self, conv, silu):
        super().__init__()
        w = conv.weight.detach()
        n, c, kh, kw = (int(v) for v in w.shape)
        from lib import _native as nat
        if w.dtype not in (torch.float16, torch.float32):
            raise nat.BeamformerError("HipConv: float16 or float32 convolutions, not %s" % w.dtype)
        self.dtype, self.f32 = w.dtype, w.dtype == torch.float32
        e = 4 if self.f32 else 8                           # elements per 16-byte chunk
        cp = 4
        while cp < c or (kw * cp) % e:
            cp *= 2
        if cp < e and (int(conv.stride[0]) % 2 or int(conv.padding[0]) % 2):
            cp = e                                          # float16: 4 channels only under the stem's even geometry
        wk = torch.zeros((n, kh, kw, cp), dtype=w.dtype, device=w.device)
        wk[..., :c] = w.permute(0, 2, 3, 1)
        row = (nat.lib.bf_conv2d_weight_row_f32 if self.f32 else nat.lib.bf_conv2d_weight_row)(kh, kw, cp)         # rows padded to whole K stages
        wp = torch.zeros((n, row), dtype=w.dtype, device=w.device)
        wp[:, : kh * kw * cp] = wk.reshape(n, -1)
        self.wp = wp.contiguous()
        self.bias = None if conv.bias is None else conv.bias.detach().float().contiguous()
        self.c, self.cp, self.n, self.kh, self.kw = c, cp, n, kh, kw
        self.stride, self.pad, self.silu = int(conv.stride[0]), int(conv.padding[0]), 1 if silu else 0
        self._conv = nat.lib.bf_conv2d_nhwc_f32_into_device if self.f32 else nat.lib.bf_conv2d_nhwc_f16_into_device
        self._cat = nat.lib.bf_conv1x1_cat_nhwc_f32_device if self.f32 else nat.lib.bf_conv1x1_cat_nhwc_f16_device

    def forward(self, x, out=None, residual=None, x2=None, up=False):
        """out: None (a new channels_last tensor) or a channel slice [B, n, Ho, Wo] of a channels_last buffer; residual: None or a
        channels_last [B, n, Ho, Wo] tensor (or slice) added to the rounded result.  1x1 layers only: x2 -- the input is
        torch.cat((x, x2), 1) without building it; up -- x is upsampled 2x (nearest) first; x / x2 may be channel slices."""
        from lib import _native as nat
        b, c, h, w = (int(v) for v in x.shape)
        stream = torch.cuda.current_stream().cuda_stream
        bias = None if self.bias is None else self.bias.data_ptr()
        if x.dtype != self.dtype:
            raise nat.BeamformerError("HipConv: a %s convolution was given a %s input" % (self.dtype, x.dtype))
        cat = x2 is not None or up or (c == self.cp and self.kh == 1 and _nhwc_view(x, b, c, h, w, self.dtype) not in (None, (x.data_ptr(), c)))
        if cat:
            if (self.kh, self.kw, self.stride, self.pad) != (1, 1, 1, 0):
                raise nat.BeamformerError("HipConv: two sources / upsampling / sliced inputs are for 1x1 layers")
            ho, wo = (2 * h, 2 * w) if up else (h, w)
            c2 = 0 if x2 is None else int(x2.shape[1])
            v1, v2 = _nhwc_view(x, b, c, h, w, self.dtype), (0, 0) if x2 is None else _nhwc_view(x2, b, c2, ho, wo, self.dtype)
            if v1 is None or v2 is None or c + c2 != self.cp:
                raise nat.BeamformerError("HipConv: sources must be channels_last %s tensors (or channel slices) with %d channels in all" % (self.dtype, self.cp))
        else:
            if c not in (self.c, self.cp):
                raise nat.BeamformerError("HipConv: expects %d input channels (or padded to %d), got %d" % (self.c, self.cp, c))
            if c != self.cp:
                xp = torch.empty((b, self.cp, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
                xp[:, :c] = x
                xp[:, c:] = 0
                x = xp
            elif not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
            ho, wo = (h + 2 * self.pad - self.kh) // self.stride + 1, (w + 2 * self.pad - self.kw) // self.stride + 1
        y = torch.empty((b, self.n, ho, wo), dtype=self.dtype, device=x.device, memory_format=torch.channels_last) if out is None else out
        vy, vr = _nhwc_view(y, b, self.n, ho, wo, self.dtype), (None, 0) if residual is None else _nhwc_view(residual, b, self.n, ho, wo, self.dtype)
        if vy is None or vr is None:
            # an NHWC buffer or a channel slice of one: channels adjacent, pixels in (b, h, w) order at one pitch
            raise nat.BeamformerError("HipConv: out / residual must be [%d, %d, %d, %d] %s channel slices of channels_last buffers" % (b, self.n, ho, wo, self.dtype))
        if cat:
            rc = self._cat(v1[0], v1[1], c, 1 if up else 0, v2[0] or None, v2[1], self.wp.data_ptr(), bias, vy[0], vy[1], vr[0], vr[1], b, ho, wo, self.cp, self.n,
                           self.silu, stream)
        else:
            rc = self._conv(x.data_ptr(), self.wp.data_ptr(), bias, vy[0], vy[1], vr[0], vr[1], b, h, w, self.cp, self.n, self.kh, self.kw, self.stride, self.pad,
                            self.silu, stream)
        if rc != 0:
            nat.check()
        return y


def use_hip_convs(net):
    """Route every convolution of a fused network (float32 or float16, already on its device) through HipConv (the SiLU moves into the kernel)."""
    for m in net.modules():
        if isinstance(m, Conv) and isinstance(m.conv, nn.Conv2d):
            m.conv, m.act = HipConv(m.conv, True), nn.Identity()
    net.detect = nn.ModuleList(HipConv(d, False) for d in net.detect)
    return net


def build(nc=1, seed=0, device="cuda", half=False, conv_backend="miopen", init_gain=None):
    """Seeded random-init network, inference mode, BatchNorm folded, channels_last on the GPU; float32 (ultralytics' predict default,
    yolo_smooth_tracking.py:13-23) or float16.
    conv_backend: "miopen" (torch's convolutions) or "hip" (this library's implicit-GEMM kernels, either precision).
    init_gain: None = torch's default initialisation (activations fade with depth: the head barely sees the image); a number g draws every
    convolution weight from N(0, (g / sqrt(fan_in))^2), which with g ~ 1.7 keeps the signal alive through the SiLU layers -- the
    image-dependent head maps the box-level agreement tests need (tests/test_detector.py)."""
    torch.manual_seed(seed)
    net = YOLOv5s(nc)
    if init_gain is not None:
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.data.normal_(0.0, init_gain / fan_in ** 0.5)
    for m in net.modules():                      # give BatchNorm non-trivial statistics so that folding is exercised
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.uniform_(-0.1, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.1, 0.1)
    for d, s in zip(net.detect, STRIDES):        # the release's bias initialisation: ~8 objects per 640 px image
        b = d.bias.view(3, -1)
        b.data[:, 4] += float(torch.log(torch.tensor(8 / (640 / s) ** 2)))
        b.data[:, 5:] += float(torch.log(torch.tensor(0.6 / (nc - 0.99999))))
    net.eval().fuse()
    net = net.to(device)
    if half:
        net = net.half()
    net = net.to(memory_format=torch.channels_last)
    if conv_backend == "hip":
        net = use_hip_convs(net)
    elif conv_backend != "miopen":
        raise ValueError("conv_backend: 'miopen' or 'hip'")
    return net
