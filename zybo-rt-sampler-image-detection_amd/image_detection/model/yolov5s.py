"""YOLOv5s-shaped detector (PyTorch-ROCm) -- the network behind the reference's `ultralytics.YOLO(best.pt)`.

The reference ships neither the architecture nor the weights (`image-detection/model/*` is git-ignored and listed in
.MISSING_LARGE_BLOBS; `ultralytics` is an unpinned third-party package), so this is the published YOLOv5s v6.x graph
(depth 0.33, width 0.50: 6x6/2 stem, C3 stages 64-128-256-512, SPPF, PAN head, three detect levels at strides 8/16/32,
anchors of the COCO release) with seeded random weights and `nc` classes (1 in the reference's use: drones).
Convolutions (fp16, channels_last, BatchNorm folded) run through the library's own implicit-GEMM MFMA kernel (`HipConv`,
csrc/conv_kernels.hip: conv + bias + SiLU in one launch) or, with conv_backend="miopen", through torch / MIOpen; the head decode
and the NMS are the hand-written HIP kernels of csrc/nms_kernels.hip."""
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
        if isinstance(self.cv3.conv, HipConv):
            # both halves of the concatenation are convolution outputs: they are written straight into its buffer
            b, _, h, w = x.shape
            c_ = self.cv1.conv.n
            buf = torch.empty((b, 2 * c_, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            self.cv2(x, out=buf[:, c_:])
            y = self.cv1(x)
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
            if nat.lib.bf_sppf_pool_device(buf.data_ptr(), b, h, w, c_, torch.cuda.current_stream().cuda_stream) != 0:
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
        self.up = nn.Upsample(scale_factor=2, mode="nearest")
        self.detect = nn.ModuleList(nn.Conv2d(c, 3 * self.no, 1) for c in (128, 256, 512))

    def up_cat(self, a, b):
        """torch.cat((self.up(a), b), 1); on the HIP convolution path one kernel (bf_upsample_concat_device)."""
        cl = torch.channels_last
        if isinstance(self.h10.conv, HipConv) and a.dtype == torch.float16 and a.is_contiguous(memory_format=cl) and b.is_contiguous(memory_format=cl):
            from lib import _native as nat
            n, ca, cb, h, w = int(a.shape[0]), int(a.shape[1]), int(b.shape[1]), int(b.shape[2]), int(b.shape[3])
            out = torch.empty((n, ca + cb, h, w), dtype=a.dtype, device=a.device, memory_format=cl)
            if nat.lib.bf_upsample_concat_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, h, w, ca, cb, torch.cuda.current_stream().cuda_stream) != 0:
                nat.check()
            return out
        return torch.cat((self.up(a), b), 1)

    def conv_cat(self, conv, x, other):
        """torch.cat((conv(x), other), 1); on the HIP convolution path the convolution writes its half of the buffer itself."""
        if isinstance(conv.conv, HipConv):
            n, c1, c2, h, w = int(other.shape[0]), conv.conv.n, int(other.shape[1]), int(other.shape[2]), int(other.shape[3])
            buf = torch.empty((n, c1 + c2, h, w), dtype=other.dtype, device=other.device, memory_format=torch.channels_last)
            conv(x, out=buf[:, :c1])
            buf[:, c1:] = other
            return buf
        return torch.cat((conv(x), other), 1)

    def forward(self, x):
        p3 = self.b4(self.b3(self.b2(self.b1(self.b0(x)))))
        p4 = self.b6(self.b5(p3))
        p5 = self.b9(self.b8(self.b7(p4)))
        t10 = self.h10(p5)
        t14 = self.h14(self.h13(self.up_cat(t10, p4)))
        o3 = self.h17(self.up_cat(t14, p3))
        o4 = self.h20(self.conv_cat(self.h18, o3, t14))
        o5 = self.h23(self.conv_cat(self.h21, o4, t10))
        return [d(o) for d, o in zip(self.detect, (o3, o4, o5))]

    def fuse(self):
        for m in self.modules():
            if isinstance(m, Conv) and isinstance(m.bn, nn.BatchNorm2d):
                m.fuse()
        return self


class HipConv(nn.Module):
    """A folded convolution (+ SiLU) through the library's own kernel: bf_conv2d_nhwc_f16_device (csrc/conv_kernels.hip, implicit
    GEMM on the f16 matrix cores).  In and out: [B, C, H, W] float16 tensors in channels_last memory, i.e. NHWC buffers.
    The weights are repacked once to [N][KH][KW][C'] with C' = the input channels padded to what the kernel takes (the 3-channel
    image of the 6x6 stem becomes 4 channels, the rest are powers of two already), each row zero-padded to whole K stages."""

    def __init__(self, conv, silu):
        super().__init__()
        w = conv.weight.detach()
        n, c, kh, kw = (int(v) for v in w.shape)
        from lib import _native as nat
        cp = 4
        while cp < c or (kw * cp) % 8:
            cp *= 2
        if cp == 4 and (int(conv.stride[0]) % 2 or int(conv.padding[0]) % 2):
            cp = 8                                          # 4 channels only under the stem's even geometry (bf_conv2d_nhwc_f16_device)
        wk = torch.zeros((n, kh, kw, cp), dtype=torch.float16, device=w.device)
        wk[..., :c] = w.permute(0, 2, 3, 1).to(torch.float16)
        wp = torch.zeros((n, nat.lib.bf_conv2d_weight_row(kh, kw, cp)), dtype=torch.float16, device=w.device)     # rows padded to whole K stages
        wp[:, : kh * kw * cp] = wk.reshape(n, -1)
        self.register_buffer("wp", wp.contiguous())
        self.register_buffer("bias", None if conv.bias is None else conv.bias.detach().float().contiguous())
        self.c, self.cp, self.n, self.kh, self.kw = c, cp, n, kh, kw
        self.stride, self.pad, self.silu = int(conv.stride[0]), int(conv.padding[0]), 1 if silu else 0

    def forward(self, x, out=None, residual=None):
        """out: None (a new channels_last tensor) or a channel slice [B, n, Ho, Wo] of a channels_last buffer; residual: None or a
        channels_last [B, n, Ho, Wo] tensor (or slice) added to the rounded result."""
        from lib import _native as nat
        b, c, h, w = (int(v) for v in x.shape)
        if x.dtype != torch.float16 or c not in (self.c, self.cp):
            raise nat.BeamformerError("HipConv: expects float16 input with %d channels (or padded to %d), got %s with %d" % (self.c, self.cp, x.dtype, c))
        if c != self.cp:
            xp = torch.empty((b, self.cp, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            xp[:, :c] = x
            xp[:, c:] = 0
            x = xp
        elif not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        ho, wo = (h + 2 * self.pad - self.kh) // self.stride + 1, (w + 2 * self.pad - self.kw) // self.stride + 1
        y = torch.empty((b, self.n, ho, wo), dtype=torch.float16, device=x.device, memory_format=torch.channels_last) if out is None else out
        for t in (y, residual):
            # an NHWC buffer or a channel slice of one: channels adjacent, pixels in (b, h, w) order at one pitch
            if t is not None and (tuple(t.shape) != (b, self.n, ho, wo) or t.dtype != torch.float16 or t.stride(1) != 1 or t.stride(2) != wo * t.stride(3) or
                                  t.stride(0) != ho * wo * t.stride(3)):
                raise nat.BeamformerError("HipConv: out / residual must be [%d, %d, %d, %d] float16 channel slices of channels_last buffers" % (b, self.n, ho, wo))
        if nat.lib.bf_conv2d_nhwc_f16_into_device(x.data_ptr(), self.wp.data_ptr(), None if self.bias is None else self.bias.data_ptr(), y.data_ptr(),
                                                  int(y.stride(3)), None if residual is None else residual.data_ptr(),
                                                  0 if residual is None else int(residual.stride(3)), b, h, w, self.cp, self.n, self.kh, self.kw,
                                                  self.stride, self.pad, self.silu, torch.cuda.current_stream().cuda_stream) != 0:
            nat.check()
        return y


def use_hip_convs(net):
    """Route every convolution of a fused, half-precision network through HipConv (the SiLU moves into the kernel)."""
    for m in net.modules():
        if isinstance(m, Conv) and isinstance(m.conv, nn.Conv2d):
            m.conv, m.act = HipConv(m.conv, True), nn.Identity()
    net.detect = nn.ModuleList(HipConv(d, False) for d in net.detect)
    return net


def build(nc=1, seed=0, device="cuda", half=True, conv_backend="miopen"):
    """Seeded random-init network, inference mode, BatchNorm folded, fp16 channels_last on the GPU.
    conv_backend: "miopen" (torch's convolutions) or "hip" (this library's implicit-GEMM kernel; float16 only)."""
    torch.manual_seed(seed)
    net = YOLOv5s(nc)
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
        if not half:
            raise ValueError("the HIP convolution kernel computes in float16: build(half=True)")
        net = use_hip_convs(net)
    elif conv_backend != "miopen":
        raise ValueError("conv_backend: 'miopen' or 'hip'")
    return net
