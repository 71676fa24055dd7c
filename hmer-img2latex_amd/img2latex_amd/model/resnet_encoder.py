"""ResNetEncoder with the reference's constructor / state_dict surface (encoder.py:132-249), HIP forward.

The reference wraps a torchvision ResNet (minus ``fc``) in ``nn.Sequential`` (encoder.py:198-199),
so its state_dict keys are ``resnet.0.weight`` (conv1), ``resnet.1.*`` (bn1), ``resnet.{4..7}.<block>.
{conv1,bn1,conv2,bn2,conv3,bn3,downsample.0,downsample.1}.*`` and ``embedding_layer.*``.  torchvision is
not a dependency here: the containers below reproduce that module tree (ResNet v1.5: stride on the 3x3
of a Bottleneck) so a reference checkpoint loads with ``strict=True``.  The reference initialises from
remote ImageNet weights (encoder.py:185-194); offline we use torchvision's random initialisation and
expect real weights to arrive through ``load_state_dict``.

Inference only: BatchNorm uses its running statistics, folded into the conv epilogue; convolutions run
in bf16 on the matrix cores with fp32 accumulation (BASELINE config 5).  Training through the ResNet
trunk (batch statistics, backward) is not built.
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
        self._ws: Optional[torch.Tensor] = None
        self._packed = {}          # id(conv) -> (version key, packed bf16 filter + folded BatchNorm)
        self.cache_packed_weights = True
        self.kernel_flags = 0      # _lib.FLAG_RESNET_NO_RING / FLAG_RESNET_IM2COL_STEM / flag_resnet_ring_depth(n)
        self.trace = None          # a list here receives (conv, bn, x, residual, y, relu, nchw_f32) of every launch

    # ------------------------------------------------------------------
    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
        return self._ws

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

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) fp32 -> (B, 512|2048) fp32: the torchvision trunk incl. global average pooling."""
        if self.training:
            raise NotImplementedError("img2latex_amd ResNetEncoder: inference only (call .eval()); BatchNorm batch "
                                      "statistics and the trunk's backward are not built")
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
        for li in range(4, 8):
            for blk in m[li]:
                identity, ishp = h, shp
                if isinstance(blk, _Bottleneck):
                    o, s1 = self._conv_bn(h, shp, blk.conv1, blk.bn1, relu=True)
                    o, s2 = self._conv_bn(o, s1, blk.conv2, blk.bn2, relu=True)
                    if blk.downsample is not None:
                        identity, _ = self._conv_bn(h, shp, blk.downsample[0], blk.downsample[1], relu=False)
                    h, shp = self._conv_bn(o, s2, blk.conv3, blk.bn3, relu=True, residual=identity)
                else:
                    o, s1 = self._conv_bn(h, shp, blk.conv1, blk.bn1, relu=True)
                    if blk.downsample is not None:
                        identity, _ = self._conv_bn(h, shp, blk.downsample[0], blk.downsample[1], relu=False)
                    h, shp = self._conv_bn(o, s1, blk.conv2, blk.bn2, relu=True, residual=identity)
            _lib.mark(f"layer{li - 3}")
        Bf, Hf, Wf, Cf = shp
        feat = torch.empty((B, Cf), dtype=torch.float32, device=x.device)
        _lib.check(L.i2l_global_avgpool_bf16_fwd(h.data_ptr(), feat.data_ptr(), B, Hf, Wf, Cf, _lib.stream_ptr()), "avgpool")
        return feat

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,3,H,W) -> (B,E)   (encoder.py:231-249)."""
        feat = self.trunk(x)
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
