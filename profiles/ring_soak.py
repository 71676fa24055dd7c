"""Race screen for the ring-buffered bf16 GEMM: every ResNet-50 layer shape, 30 launches each, outputs must be
bit-identical from launch to launch (the kernel is deterministic; a read before its DMA landed would show up as a
difference) and equal to the single-buffered kernel's result."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib  # noqa: E402
from resnet_layers import LAYERS  # noqa: E402


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B = 64
    bad = 0
    for name, H, W, Cin, Cout, k, s, pd, res, cnt in LAYERS:
        Ho, Wo = (H + 2 * pd - k) // s + 1, (W + 2 * pd - k) // s + 1
        g = torch.Generator(device="cpu").manual_seed(hash(name) % 1000)
        x = (torch.randn(B, H, W, Cin, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5).to(dev)
        ones, zeros = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        nb = L.i2l_conv_bf16_packed_bytes(Cout, Cin, k, k)
        packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(L.i2l_conv_bn_bf16_pack(w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), ones.data_ptr(), 1e-5,
                                           packed.data_ptr(), nb, Cout, Cin, k, k, _lib.stream_ptr()), "pack")
        r = torch.randn(B, Ho, Wo, Cout, generator=g).to(torch.bfloat16).to(dev) if res else None
        wsb = L.i2l_conv_bf16_workspace_bytes(B, H, W, Cin, Cout, k, k, s, pd, 0)
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
        ref = None
        for rep in range(30):
            y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
            _lib.check(L.i2l_conv_bn_act_bf16_fwd(x.data_ptr(), 0, packed.data_ptr(), _lib.ptr(r), y.data_ptr(), B, H, W, Cin, Cout,
                                                  k, k, s, pd, 1, ws.data_ptr(), wsb, 0, _lib.stream_ptr()), "conv")
            yi = y.view(torch.int16)
            if ref is None:
                ref = yi.clone()
            elif not torch.equal(ref, yi):
                bad += 1
                print("DIFF", name, rep, int((ref != yi).sum()), flush=True)
        print("ok", name, flush=True)
    print("bad =", bad)


if __name__ == "__main__":
    main()
