"""CNNEncoder with the reference's constructor / state_dict surface, HIP forward.

Reference: img2latex/model/encoder.py:16-129.  The nn.Conv2d / nn.Linear children
exist only to own the parameters under the reference's state_dict keys
(``cnn_layers.{0,3,6}.{weight,bias}``, ``embedding_layer.{weight,bias}``) and to get
PyTorch's default initialisation in the reference's creation order; their own
``forward`` is never called -- the math runs in libimg2latex_hip.so.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib


class CNNEncoder(nn.Module):
    def __init__(self, img_height: int = None, img_width: int = None, channels: int = None,
                 conv_filters: List[int] = None, kernel_size: int = None, pool_size: int = None,
                 padding: str = "same", embedding_dim: int = None):
        super().__init__()
        # defaults of encoder.py:51-64
        img_height = 64 if img_height is None else img_height
        img_width = 800 if img_width is None else img_width
        channels = 1 if channels is None else channels
        conv_filters = [32, 64, 128] if conv_filters is None else list(conv_filters)
        kernel_size = 3 if kernel_size is None else kernel_size
        pool_size = 2 if pool_size is None else pool_size
        embedding_dim = 256 if embedding_dim is None else embedding_dim
        if kernel_size != 3 or pool_size != 2 or padding != "same":
            raise NotImplementedError(
                "img2latex_amd CNNEncoder: the HIP block kernel is conv3x3(pad 1)+ReLU+maxpool2 "
                f"(every reference config uses it); got kernel_size={kernel_size}, pool_size={pool_size}, "
                f"padding={padding!r}")
        self.img_height, self.img_width = img_height, img_width
        self.channels, self.embedding_dim = channels, embedding_dim
        self.conv_filters = conv_filters

        layers: List[nn.Module] = []
        cin, h, w = channels, img_height, img_width
        for cout in conv_filters:                       # Sequential idx 0,3,6 = conv (encoder.py:78-95)
            layers += [nn.Conv2d(cin, cout, kernel_size, padding=kernel_size // 2), nn.ReLU(),
                       nn.MaxPool2d(kernel_size=pool_size)]
            cin, h, w = cout, h // pool_size, w // pool_size
        self.cnn_layers = nn.Sequential(*layers)
        self.flatten = nn.Flatten()
        self.flattened_size = cin * h * w              # what the reference finds by a dummy forward (:99-102)
        self.embedding_layer = nn.Linear(self.flattened_size, embedding_dim)
        self.activation = nn.ReLU()
        # Workspaces are per (device, STREAM): GreedyPipeline(encoder_streams=2) runs two forwards of this module side by
        # side, and the FC layer's split-K slabs / a block's packed filter image must not be shared between them
        self._ws = {}              # (device, stream) -> scratch
        self._packed_ws = {}       # (block index, stream) -> (workspace holding the packed filters, key they were packed for)
        self.kernel_flags = 0      # _lib.FLAG_EXACT_FP32: exact fp32 kernels instead of the 3 x bf16 split

    def _workspace(self, nbytes: int, device) -> Optional[torch.Tensor]:
        if nbytes == 0:
            return None
        key = (str(device), _lib.stream_ptr())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return ws

    def conv_blocks(self, x: torch.Tensor, argmax_out: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
        """Outputs of the conv blocks (each = Conv2d+ReLU+MaxPool2d fused in one launch).
        ``argmax_out``: a list that receives the uint8 pooling-argmax maps (training forward)."""
        x = _lib.require_gpu(x, "images")
        if x.dim() != 4 or x.shape[1] != self.channels:
            raise RuntimeError(f"expected (B,{self.channels},H,W) images, got {tuple(x.shape)}")
        L = _lib.lib()
        outs = []
        _lib.mark("begin")
        for i in range(len(self.conv_filters)):
            conv = self.cnn_layers[3 * i]
            B, cin, h, w = x.shape
            y = torch.empty((B, conv.out_channels, h // 2, w // 2), dtype=torch.float32, device=x.device)
            wt = _lib.require_gpu(conv.weight.detach(), "conv weight")
            bs = _lib.require_gpu(conv.bias.detach(), "conv bias")
            nbytes = L.i2l_conv_workspace_bytes(cin, conv.out_channels)
            amax = None
            flags = self.kernel_flags
            if argmax_out is not None:
                amax = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
                argmax_out.append(amax)
                ws = self._workspace(nbytes, x.device)
            else:
                # inference: every block keeps its own workspace, so the packed filter image written there survives from
                # batch to batch; it is rebuilt only when the weight changed ((data_ptr, _version), as LSTMDecoder.prepare)
                key = (wt.data_ptr(), conv.weight._version, x.device, nbytes, self.kernel_flags)
                slot = (i, _lib.stream_ptr())
                held = self._packed_ws.get(slot)
                if held is None or held[0].numel() < max(nbytes, 16) or held[0].device != x.device:
                    held = (torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device), None)
                ws = held[0] if nbytes else None
                if held[1] == key:
                    flags |= _lib.FLAG_WEIGHTS_PACKED
                self._packed_ws[slot] = (held[0], key)
            _lib.check(L.i2l_conv3x3_relu_pool2_fwd(x.data_ptr(), wt.data_ptr(), bs.data_ptr(), y.data_ptr(),
                                                    _lib.ptr(amax), B, cin, h, w, conv.out_channels, _lib.ptr(ws),
                                                    nbytes, flags, _lib.stream_ptr()),
                       "conv3x3_relu_pool2_fwd")
            _lib.mark(f"conv{i}")
            outs.append(y)
            x = y
        return outs

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,H,W) -> (B,E)   (encoder.py:111-129).  With gradients enabled the result carries a
        grad_fn whose backward runs the HIP gradient kernels."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from ._train_fn import CNNEncoderFn
            return CNNEncoderFn.apply(self, x, *self.parameters())
        return self._forward_impl(x)

    def _linear(self, feat: torch.Tensor) -> torch.Tensor:
        B = feat.shape[0]
        K = feat.numel() // B                           # Flatten: NCHW row-major, a view
        if K != self.embedding_layer.in_features:
            raise RuntimeError(f"flattened size {K} != embedding_layer.in_features "
                               f"{self.embedding_layer.in_features} (image size differs from the constructor's)")
        E = self.embedding_dim
        L = _lib.lib()
        out = torch.empty((B, E), dtype=torch.float32, device=feat.device)
        nbytes = L.i2l_linear_workspace_bytes(B, K, E)
        ws = self._workspace(nbytes, feat.device)
        wt = _lib.require_gpu(self.embedding_layer.weight.detach(), "embedding_layer.weight")
        bs = _lib.require_gpu(self.embedding_layer.bias.detach(), "embedding_layer.bias")
        _lib.check(L.i2l_linear_bias_act_fwd(feat.data_ptr(), wt.data_ptr(), bs.data_ptr(), out.data_ptr(),
                                             B, K, E, 1, _lib.ptr(ws), nbytes, self.kernel_flags, _lib.stream_ptr()),
                   "linear_bias_act_fwd")
        _lib.mark("fc")
        return out

    def _forward_impl(self, x: torch.Tensor, argmax_out: Optional[List[torch.Tensor]] = None,
                      blocks_out: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        blocks = self.conv_blocks(x, argmax_out)
        if blocks_out is not None:
            blocks_out.extend(blocks)
        return self._linear(blocks[-1])
