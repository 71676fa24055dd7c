"""ResNetEncoder with the reference's constructor / state_dict surface (encoder.py:132-249), HIP forward.

The reference wraps a torchvision ResNet (minus ``fc``) in ``nn.Sequential`` (encoder.py:198-199),
so its state_dict keys are ``resnet.0.weight`` (conv1), ``resnet.1.*`` (bn1), ``resnet.{4..7}.<block>.
{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}.*`` and ``embedding_layer.*``.  torchvision is
not a dependency here: the containers below reproduce that module tree (ResNet v1.5: stride on the 3x3
of a Bottleneck) so a reference checkpoint loads with ``strict=True``.  The reference initialises from
remote ImageNet weights (encoder.py:185-194); offline we use torchvision's random initialisation and
expect real weights to arrive through ``load_state_dict``.

Inference (``.eval()``): BatchNorm uses its running statistics, folded into the conv epilogue; convolutions run
in bf16 on the matrix cores with fp32 accumulation (BASELINE config 5).

Training (``.train()``, r03; fp32 grade since r04): every BatchNorm2d normalises with BATCH statistics and updates its
running statistics (also the frozen ones: ``freeze_backbone`` only clears ``requires_grad``, encoder.py:201-210); the
forward keeps a tape (raw conv outputs, batch statistics, activations -- all NHWC fp32, the convolutions on the
split-bf16 matrix-core GEMM, i.e. the arithmetic of the reference's fp32 branch) and the backward walks it in reverse
through the HIP gradient kernels (csrc/resnet_train.hip): BatchNorm / ReLU / residual backward, conv weight and data
gradients as GEMMs over an explicit im2col image, max-pool and average-pool backward.  Gradients stop at the first unit that has a trainable
parameter: with the default ``freeze_backbone=True`` that is layer4.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib

_SPECS = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]),
          "resnet50": ("bottleneck", [3, 4, 6, 3]), "resnet101": ("bottleneck", [3, 4, 23, 3]),
          "resnet152": ("bottleneck", [3, 8, 36, 3])}


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 3, stride), nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        self.downsample = downsample


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample


def _make_layer(block, inplanes, planes, n, stride):
    layers = []
    for i in range(n):
        s = stride if i == 0 else 1
        down = None
        if s != 1 or inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, stride=s, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers.append(block(inplanes, planes, s, down))
        inplanes = planes * block.expansion
    return nn.Sequential(*layers), inplanes


class ResNetEncoder(nn.Module):
    # kernel flags a pipeline ORs in when SEVERAL trunks are in flight (GreedyPipeline(encoder_streams >= 2)): the tile shapes
    # that balance one layer over the chip only cost arithmetic intensity when a neighbour fills the idle CUs (r04, measured)
    MULTI_STREAM_FLAGS = _lib.FLAG_RESNET_WIDE_TILES | _lib.FLAG_RESNET_NO_PATCH

    def __init__(self, img_height: int = None, img_width: int = None, channels: int = None,
                 model_name: str = "resnet50", embedding_dim: int = None, freeze_backbone: bool = True):
        super().__init__()
        img_height = 64 if img_height is None else img_height          # defaults of encoder.py:163-170
        img_width = 800 if img_width is None else img_width
        channels = 3 if channels is None else channels
        embedding_dim = 256 if embedding_dim is None else embedding_dim
        self.img_height, self.img_width, self.channels, self.embedding_dim = img_height, img_width, channels, embedding_dim
        if model_name not in _SPECS:
            raise ValueError(f"Invalid ResNet model name: {model_name}")          # encoder.py:196
        kind, counts = _SPECS[model_name]
        block = _BasicBlock if kind == "basic" else _Bottleneck
        modules: List[nn.Module] = [nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
                                    nn.ReLU(inplace=True), nn.MaxPool2d(3, stride=2, padding=1)]
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), counts)):
            layer, inplanes = _make_layer(block, inplanes, planes, n, 1 if i == 0 else 2)
            modules.append(layer)
        modules.append(nn.AdaptiveAvgPool2d((1, 1)))
        self.resnet = nn.Sequential(*modules)                                      # children minus fc (:198-199)
        for mod in self.resnet.modules():                                          # torchvision's random init
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
        if freeze_backbone:                                                        # :201-210
            for p in self.resnet.parameters():
                p.requires_grad = False
            for p in modules[-2].parameters():                                     # layer4 stays trainable
                p.requires_grad = True
        self.flatten = nn.Flatten()
        self.embedding_layer = nn.Linear(512 if kind == "basic" else 2048, embedding_dim)
        self.activation = nn.ReLU()
        self.model_name = model_name
        self._ws = None            # {(device, stream): workspace bytes}
        self._packed = {}          # id(conv) -> (version key, packed bf16 filter + folded BatchNorm)
        self.cache_packed_weights = True
        self.kernel_flags = 0      # _lib.FLAG_RESNET_NO_RING / FLAG_RESNET_IM2COL_STEM / flag_resnet_ring_depth(n)
        self.trace = None          # a list here receives (conv, bn, x, residual, y, relu, nchw_f32) of every launch
        self.joined_heads = []     # with trace: the conv1 modules whose input came from a join (never read back from HBM)
        # layer1: conv3 + identity + ReLU and the next block's conv1 in one launch (i2l_bottleneck_join_bf16_fwd, r04).  OFF by
        # default: bit-identical, saves a 168 MB read per block, and is still SLOWER than the two launches (117 - 134 us against
        # 103 - 125 at B=256: one workgroup per CU walks its tiles phase by phase; profiles/r04/resnet_patch.txt, 8)
        self.fuse_joins = False
        self.trace_bwd = None      # a list here receives every unit's backward: dict(unit, dy, dz, dx, dres, names)

    # ------------------------------------------------------------------
    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        key = (str(device), _lib.stream_ptr())              # one workspace per stream: pipelines run encoders side by side
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
        return ws

    def _conv_bn(self, x, shape, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool, residual=None, nchw_f32=False):
        """One fused conv+BN(+residual)(+ReLU) launch; x is NHWC bf16 (or the NCHW fp32 images)."""
        B, H, W, Cin = shape
        k, s, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (H + 2 * pd - k) // s + 1, (W + 2 * pd - k) // s + 1
        L = _lib.lib()
        packed = self._packed_weights(conv, bn, x.device)
        y = torch.empty((B, Ho, Wo, conv.out_channels), dtype=torch.bfloat16, device=x.device)
        nbytes = L.i2l_conv_bf16_workspace_bytes(B, H, W, Cin, conv.out_channels, k, k, s, pd, self.kernel_flags)
        ws = self._workspace(nbytes, x.device)
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(
            x.data_ptr(), 1 if nchw_f32 else 0, packed.data_ptr(), _lib.ptr(residual), y.data_ptr(), B, H, W, Cin,
            conv.out_channels, k, k, s, pd, 1 if relu else 0, ws.data_ptr(), nbytes, self.kernel_flags,
            _lib.stream_ptr()), "conv_bn_act_bf16_fwd")
        if self.trace is not None:
            self.trace.append((conv, bn, x, residual, y, relu, nchw_f32))
        return y, (B, Ho, Wo, conv.out_channels)

    def _packed_weights(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, device) -> torch.Tensor:
        """bf16 filter image + BatchNorm folded into scale/bias (i2l_conv_bn_bf16_pack): weight-only work, redone
        whenever a weight or a BatchNorm statistic changed (or on every call with cache_packed_weights = False)."""
        tensors = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._packed.get(id(conv))
        if self.cache_packed_weights and hit is not None and hit[0] == key and hit[1].device == device:
            return hit[1]
        L = _lib.lib()
        k = conv.kernel_size[0]
        nbytes = L.i2l_conv_bf16_packed_bytes(conv.out_channels, conv.in_channels, k, k)
        buf = hit[1] if hit is not None and hit[1].numel() >= nbytes and hit[1].device == device else \
            torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(L.i2l_conv_bn_bf16_pack(conv.weight.detach().data_ptr(), bn.weight.detach().data_ptr(),
                                           bn.bias.detach().data_ptr(), bn.running_mean.data_ptr(),
                                           bn.running_var.data_ptr(), float(bn.eps), buf.data_ptr(), nbytes,
                                           conv.out_channels, conv.in_channels, k, k, _lib.stream_ptr()),
                   "conv_bn_bf16_pack")
        self._packed[id(conv)] = (key, buf)
        return buf

    def _join(self, o, shape, blk, identity, nxt):
        """conv3 + bn3 + identity + ReLU of `blk` and conv1 + bn1 + ReLU of the next block in one launch
        (i2l_bottleneck_join_bf16_fwd) where the block output's tile fits in LDS: 64 -> 256 -> 64 | 128 channels, i.e. layer1 of
        the bottleneck ResNets and its hand-over to layer2.  Returns (y, y_shape, (z, z_shape)) or None (two launches)."""
        if nxt is None or not isinstance(nxt, _Bottleneck):
            return None
        c3, c1 = blk.conv3, nxt.conv1
        if (c3.in_channels, c3.out_channels) != (64, 256) or c1.in_channels != 256 or c1.out_channels not in (64, 128):
            return None
        if c3.stride[0] != 1 or c1.stride[0] != 1 or c1.kernel_size[0] != 1:
            return None
        B, H, W, _ = shape
        L = _lib.lib()
        y = torch.empty((B, H, W, 256), dtype=torch.bfloat16, device=o.device)
        z = torch.empty((B, H, W, c1.out_channels), dtype=torch.bfloat16, device=o.device)
        _lib.check(L.i2l_bottleneck_join_bf16_fwd(
            o.data_ptr(), self._packed_weights(c3, blk.bn3, o.device).data_ptr(), identity.data_ptr(), y.data_ptr(),
            self._packed_weights(c1, nxt.bn1, o.device).data_ptr(), z.data_ptr(), B * H * W, 64, 256, c1.out_channels,
            _lib.stream_ptr()), "bottleneck_join_bf16_fwd")
        if self.trace is not None:                    # the same two records the two launches would leave
            self.trace.append((c3, blk.bn3, o, identity, y, True, False))
            self.trace.append((c1, nxt.bn1, y, None, z, True, False))
            self.joined_heads.append(c1)              # ... whose input was not read back from HBM
        return y, (B, H, W, 256), (z, (B, H, W, c1.out_channels))

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) fp32 -> (B, 512|2048) fp32: the torchvision trunk incl. global average pooling."""
        if self.training:
            return self._trunk_train(x, None)
        x = _lib.require_gpu(x, "images")
        B, C, H, W = x.shape
        if C != 3:
            raise RuntimeError(f"ResNet expects 3-channel images, got {C}")
        L = _lib.lib()
        m = self.resnet
        _lib.mark("begin")
        h, shp = self._conv_bn(x, (B, H, W, 3), m[0], m[1], relu=True, nchw_f32=True)
        Bh, Hh, Wh, Ch = shp
        Ho, Wo = (Hh - 1) // 2 + 1, (Wh - 1) // 2 + 1
        p = torch.empty((B, Ho, Wo, Ch), dtype=torch.bfloat16, device=x.device)
        _lib.check(L.i2l_maxpool3x3s2_bf16_fwd(h.data_ptr(), p.data_ptr(), B, Hh, Wh, Ch, _lib.stream_ptr()), "maxpool")
        h, shp = p, (B, Ho, Wo, Ch)
        blocks = [blk for li in range(4, 8) for blk in m[li]]
        layer_end = {id(m[li][-1]): li - 3 for li in range(4, 8)}
        head = None                                   # (o, shape): this block's conv1 output, already made by the previous block's join
        for bi, blk in enumerate(blocks):
            identity, ishp = h, shp
            if isinstance(blk, _Bottleneck):
                if head is not None:
                    (o, s1), head = head, None
                else:
                    o, s1 = self._conv_bn(h, shp, blk.conv1, blk.bn1, relu=True)
                o, s2 = self._conv_bn(o, s1, blk.conv2, blk.bn2, relu=True)
                if blk.downsample is not None:
                    identity, _ = self._conv_bn(h, shp, blk.downsample[0], blk.downsample[1], relu=False)
                nxt = blocks[bi + 1] if bi + 1 < len(blocks) else None
                joined = self._join(o, s2, blk, identity, nxt) if self.fuse_joins else None
                if joined is not None:
                    h, shp, head = joined
                else:
                    h, shp = self._conv_bn(o, s2, blk.conv3, blk.bn3, relu=True, residual=identity)
            else:
                o, s1 = self._conv_bn(h, shp, blk.conv1, blk.bn1, relu=True)
                if blk.downsample is not None:
                    identity, _ = self._conv_bn(h, shp, blk.downsample[0], blk.downsample[1], relu=False)
                h, shp = self._conv_bn(o, s1, blk.conv2, blk.bn2, relu=True, residual=identity)
            if id(blk) in layer_end:
                _lib.mark(f"layer{layer_end[id(blk)]}")
        Bf, Hf, Wf, Cf = shp
        feat = torch.empty((B, Cf), dtype=torch.float32, device=x.device)
        _lib.check(L.i2l_global_avgpool_bf16_fwd(h.data_ptr(), feat.data_ptr(), B, Hf, Wf, Cf, _lib.stream_ptr()), "avgpool")
        return feat

    # ------------------------------------------------------------------ training mode (batch statistics, backward)
    def _conv_bn_train(self, x, shape, conv, bn, relu, residual, nchw, tape):
        """conv -> z, batch-statistics BatchNorm (+ residual) (+ ReLU) -> y, all NHWC fp32 (fp32-grade GEMM on the
        split-bf16 matrix-core kernel, csrc/resnet_train.hip); records the unit on the tape."""
        B, H, W, Cin = shape
        k, s, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Ho, Wo = (H + 2 * pd - k) // s + 1, (W + 2 * pd - k) // s + 1
        L = _lib.lib()
        dev, co = x.device, conv.out_channels
        kind = 2 if nchw else 1
        z = torch.empty((B, Ho, Wo, co), dtype=torch.float32, device=dev)
        # a unit whose weight gradient will be asked for keeps its OWN workspace on the tape: the fp32 column image the
        # forward GEMM read is what the weight-gradient GEMM reads again (I2L_FLAG_CONV_COL_READY: no second im2col)
        direct = k == 1 and s == 1 and pd == 0 and not nchw
        keep_col = tape is not None and conv.weight.requires_grad and not direct
        nbytes = L.i2l_conv_f32_workspace_bytes(kind, B, H, W, Cin, co, k, k, s, pd, 1 if keep_col and not nchw else 0)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if keep_col else self._workspace(nbytes, dev)
        conv_ws = ws if keep_col else None
        _lib.check(L.i2l_conv_f32_fwd(x.data_ptr(), kind, conv.weight.detach().data_ptr(), z.data_ptr(), B, H, W, Cin, co,
                                      k, k, s, pd, ws.data_ptr(), nbytes, self.kernel_flags & _lib.FLAG_EXACT_FP32,
                                      _lib.stream_ptr()), "conv_f32_fwd")
        M = B * Ho * Wo
        y = torch.empty_like(z)
        mean = torch.empty((co,), dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        nb = L.i2l_bn_train_workspace_bytes(M, co)
        ws = self._workspace(nb, dev)
        track = bn.track_running_stats and bn.running_mean is not None
        momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        if track:
            bn.num_batches_tracked.add_(1)
            if bn.momentum is None:                                  # cumulative moving average (nn.BatchNorm2d)
                momentum = 1.0 / float(bn.num_batches_tracked)
        _lib.check(L.i2l_bn_train_fwd_f32(z.data_ptr(), _lib.ptr(residual), bn.weight.detach().data_ptr(),
                                          bn.bias.detach().data_ptr(), bn.running_mean.data_ptr() if track else None,
                                          bn.running_var.data_ptr() if track else None, momentum, float(bn.eps),
                                          1 if relu else 0, y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), M, co,
                                          ws.data_ptr(), nb, _lib.stream_ptr()), "bn_train_fwd_f32")
        if track:
            for t in (bn.running_mean, bn.running_var):              # written through raw pointers: packed-weight caches
                torch.autograd.graph.increment_version(t)
        if tape is not None:
            tape.append(dict(conv=conv, bn=bn, x=x, in_shape=shape, nchw=nchw, z=z, y=y, relu=relu, mean=mean,
                             invstd=invstd, out_shape=(B, Ho, Wo, co), residual=residual, conv_ws=conv_ws))
        return y, (B, Ho, Wo, co)

    def _trunk_train(self, x: torch.Tensor, tape) -> torch.Tensor:
        """The trunk under model.train().  ``tape``: a dict that receives what the backward needs (None: forward only)."""
        x = _lib.require_gpu(x, "images")
        B, C, H, W = x.shape
        if C != 3:
            raise RuntimeError(f"ResNet expects 3-channel images, got {C}")
        L = _lib.lib()
        m = self.resnet
        units = [] if tape is not None else None
        h, shp = self._conv_bn_train(x, (B, H, W, 3), m[0], m[1], True, None, True, units)
        Bh, Hh, Wh, Ch = shp
        Ho, Wo = (Hh - 1) // 2 + 1, (Wh - 1) // 2 + 1
        p = torch.empty((B, Ho, Wo, Ch), dtype=torch.float32, device=x.device)
        _lib.check(L.i2l_maxpool3x3s2_f32_fwd(h.data_ptr(), p.data_ptr(), B, Hh, Wh, Ch, _lib.stream_ptr()), "maxpool")
        pool_in, pool_shape = h, shp
        h, shp = p, (B, Ho, Wo, Ch)
        blocks = []
        for li in range(4, 8):
            for blk in m[li]:
                identity, rec = h, dict(kind="bottleneck" if isinstance(blk, _Bottleneck) else "basic", down=None)
                first = len(units) if units is not None else 0
                if isinstance(blk, _Bottleneck):
                    o, s1 = self._conv_bn_train(h, shp, blk.conv1, blk.bn1, True, None, False, units)
                    o, s2 = self._conv_bn_train(o, s1, blk.conv2, blk.bn2, True, None, False, units)
                    if blk.downsample is not None:
                        identity, _ = self._conv_bn_train(h, shp, blk.downsample[0], blk.downsample[1], False, None, False, units)
                        rec["down"] = len(units) - 1 if units is not None else None
                    h, shp = self._conv_bn_train(o, s2, blk.conv3, blk.bn3, True, identity, False, units)
                else:
                    o, s1 = self._conv_bn_train(h, shp, blk.conv1, blk.bn1, True, None, False, units)
                    if blk.downsample is not None:
                        identity, _ = self._conv_bn_train(h, shp, blk.downsample[0], blk.downsample[1], False, None, False, units)
                        rec["down"] = len(units) - 1 if units is not None else None
                    h, shp = self._conv_bn_train(o, s1, blk.conv2, blk.bn2, True, identity, False, units)
                if units is not None:
                    rec["main"] = [i for i in range(first, len(units)) if i != rec["down"]]
                    blocks.append(rec)
        Bf, Hf, Wf, Cf = shp
        feat = torch.empty((B, Cf), dtype=torch.float32, device=x.device)
        _lib.check(L.i2l_global_avgpool_f32_fwd(h.data_ptr(), feat.data_ptr(), B, Hf, Wf, Cf, _lib.stream_ptr()), "avgpool")
        if tape is not None:
            tape.update(units=units, blocks=blocks, pool_in=pool_in, pool_shape=pool_shape, final_shape=shp)
        return feat

    def _conv_backward(self, u, dz: torch.Tensor, need_dx: bool, grads, name: str):
        """Weight gradient (into grads[name + '.weight'] when trainable) and data gradient of one conv from dz (M, Cout)."""
        L = _lib.lib()
        conv = u["conv"]
        B, H, W, Cin = u["in_shape"]
        k, s, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        co = u["out_shape"][3]
        dev = dz.device
        want_dw = conv.weight.requires_grad
        if not (want_dw or need_dx):
            return None
        kind = 2 if u["nchw"] else 1
        dx = torch.empty((B, H, W, Cin), dtype=torch.float32, device=dev) if need_dx else None
        nbytes = L.i2l_conv_f32_workspace_bytes(kind, B, H, W, Cin, co, k, k, s, pd, 1 if need_dx else 0)
        flags = self.kernel_flags & _lib.FLAG_EXACT_FP32
        ws = u.get("conv_ws")
        if ws is not None and want_dw and ws.numel() >= nbytes:
            flags |= _lib.FLAG_CONV_COL_READY            # the forward call's column image is still at the head of this workspace
            u["conv_ws"] = None                          # ... and after this call it no longer is (dcol / GEMM slabs behind it are)
        else:
            ws = self._workspace(nbytes, dev)
        _lib.check(L.i2l_conv_f32_bwd(u["x"].data_ptr(), kind, conv.weight.detach().data_ptr(), dz.data_ptr(), _lib.ptr(dx),
                                      grads[name + ".weight"].data_ptr() if want_dw else None, B, H, W, Cin, co, k, k, s, pd,
                                      ws.data_ptr(), nbytes, flags, _lib.stream_ptr()),
                   "conv_f32_bwd")
        return dx

    def _unit_backward(self, u, dy, need_dx, grads, names, dres=None, dres_accumulate=False):
        """BatchNorm (+ReLU) backward of one unit, then its conv's; returns dx (or None)."""
        L = _lib.lib()
        B, Ho, Wo, co = u["out_shape"]
        M = B * Ho * Wo
        bn = u["bn"]
        nm_conv, nm_bn = names[id(u["conv"])], names[id(bn)]
        dz = torch.empty((B, Ho, Wo, co), dtype=torch.float32, device=dy.device)
        nb = L.i2l_bn_train_workspace_bytes(M, co)
        ws = self._workspace(nb, dy.device)
        dgamma = grads[nm_bn + ".weight"] if bn.weight.requires_grad else None
        dbeta = grads[nm_bn + ".bias"] if bn.bias.requires_grad else None
        _lib.check(L.i2l_bn_train_bwd_f32(dy.data_ptr(), u["y"].data_ptr() if u["relu"] else None, u["z"].data_ptr(),
                                           bn.weight.detach().data_ptr(), u["mean"].data_ptr(), u["invstd"].data_ptr(),
                                           dz.data_ptr(), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dres),
                                           1 if dres_accumulate else 0, M, co, ws.data_ptr(), nb, _lib.stream_ptr()),
                   "bn_train_bwd_f32")
        dx = self._conv_backward(u, dz, need_dx, grads, nm_conv)
        if self.trace_bwd is not None:
            self.trace_bwd.append(dict(unit=u, dy=dy, dz=dz, dx=dx, dres=dres, conv_name=nm_conv, bn_name=nm_bn))
        return dx

    def _trunk_backward(self, tape, dfeat: torch.Tensor, grads) -> None:
        """d(trunk features) (B, C) -> gradients of every trainable trunk parameter (``grads``: encoder-relative names)."""
        L = _lib.lib()
        units, blocks = tape["units"], tape["blocks"]
        names = {}
        for n, mod in self.named_modules():
            names[id(mod)] = n
        trainable = [any(p.requires_grad for p in list(u["conv"].parameters()) + list(u["bn"].parameters())) for u in units]
        first_trainable = trainable.index(True) if any(trainable) else len(units)
        B, Hf, Wf, Cf = tape["final_shape"]
        dh = torch.empty((B, Hf, Wf, Cf), dtype=torch.float32, device=dfeat.device)
        _lib.check(L.i2l_global_avgpool_bwd_f32(dfeat.data_ptr(), dh.data_ptr(), B, Hf, Wf, Cf, _lib.stream_ptr()),
                   "global_avgpool_bwd")
        for rec in reversed(blocks):
            main, down = rec["main"], rec["down"]
            block_first = min(main + ([down] if down is not None else []))
            if max(main + ([down] if down is not None else [])) < first_trainable:
                return                                              # nothing upstream needs a gradient
            need_in = block_first > first_trainable                  # does the block's INPUT need a gradient?
            last = units[main[-1]]
            dres = torch.empty_like(dh)                              # gradient of the residual branch = masked dh
            d = self._unit_backward(last, dh, True, grads, names, dres=dres)
            for idx in reversed(main[:-1]):
                need = idx > block_first or need_in
                if idx == main[0]:
                    need = need_in
                d = self._unit_backward(units[idx], d, need, grads, names)
            if down is not None:
                dd = self._unit_backward(units[down], dres, need_in, grads, names)
                if need_in:
                    d = d + dd
            elif need_in:
                d = d + dres
            if not need_in:
                return
            dh = d
        # stem: max-pool, then conv1 + bn1
        if first_trainable == 0:
            Bp, Hp, Wp, Cp = tape["pool_shape"]
            dpool = torch.empty((Bp, Hp, Wp, Cp), dtype=torch.float32, device=dfeat.device)
            _lib.check(L.i2l_maxpool3x3s2_f32_bwd(tape["pool_in"].data_ptr(), dh.data_ptr(), dpool.data_ptr(), Bp, Hp, Wp, Cp,
                                                   _lib.stream_ptr()), "maxpool3x3s2_bwd")
            self._unit_backward(units[0], dpool, False, grads, names)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,E)   (encoder.py:231-249).  In training mode with gradients enabled the result carries a
        grad_fn whose backward runs the HIP gradient kernels."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if not self.training:
                raise NotImplementedError("img2latex_amd ResNetEncoder: gradients are built for training mode (batch "
                                          "statistics), as the reference's Trainer uses it; call .train(), or torch.no_grad()")
            from ._train_fn import ResNetEncoderFn
            return ResNetEncoderFn.apply(self, x, *[p for p in self.parameters() if p.requires_grad])
        return self._head(self.trunk(x))

    def _head(self, feat: torch.Tensor) -> torch.Tensor:
        B, K = feat.shape
        E = self.embedding_dim
        L = _lib.lib()
        out = torch.empty((B, E), dtype=torch.float32, device=feat.device)
        nbytes = L.i2l_linear_workspace_bytes(B, K, E)
        ws = self._workspace(nbytes, feat.device)
        _lib.check(L.i2l_linear_bias_act_fwd(feat.data_ptr(), self.embedding_layer.weight.detach().data_ptr(),
                                             self.embedding_layer.bias.detach().data_ptr(), out.data_ptr(), B, K, E, 1,
                                             ws.data_ptr(), nbytes, 0, _lib.stream_ptr()), "linear_bias_act_fwd")
        _lib.mark("fc")
        return out
