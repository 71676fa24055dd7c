"""Per-layer timing of the bf16 ResNet-50 convolutions at the bench shape (B=256, 3x64x320).

    python profiles/resnet_layers.py            # ring-buffered GEMM (default)
    FLAGS=4 python profiles/resnet_layers.py    # I2L_FLAG_RESNET_NO_RING: single-buffered kernel (FLAGS = I2L_FLAG_* bits)

Prints microseconds, TFLOP/s and the algorithmic GB/s (input + weights + residual + output, each once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib  # noqa: E402

# (name, H, W, Cin, Cout, k, stride, pad, residual, count in ResNet-50)
LAYERS = [
    ("l1.conv1a", 16, 80, 64, 64, 1, 1, 0, 0, 1), ("l1.conv2", 16, 80, 64, 64, 3, 1, 1, 0, 3),
    ("l1.down", 16, 80, 64, 256, 1, 1, 0, 0, 1), ("l1.conv3", 16, 80, 64, 256, 1, 1, 0, 1, 3),
    ("l1.conv1", 16, 80, 256, 64, 1, 1, 0, 0, 2),
    ("l2.conv1a", 16, 80, 256, 128, 1, 1, 0, 0, 1), ("l2.conv2a", 16, 80, 128, 128, 3, 2, 1, 0, 1),
    ("l2.down", 16, 80, 256, 512, 1, 2, 0, 0, 1), ("l2.conv3", 8, 40, 128, 512, 1, 1, 0, 1, 4),
    ("l2.conv1", 8, 40, 512, 128, 1, 1, 0, 0, 3), ("l2.conv2", 8, 40, 128, 128, 3, 1, 1, 0, 3),
    ("l3.conv1a", 8, 40, 512, 256, 1, 1, 0, 0, 1), ("l3.conv2a", 8, 40, 256, 256, 3, 2, 1, 0, 1),
    ("l3.down", 8, 40, 512, 1024, 1, 2, 0, 0, 1), ("l3.conv3", 4, 20, 256, 1024, 1, 1, 0, 1, 6),
    ("l3.conv1", 4, 20, 1024, 256, 1, 1, 0, 0, 5), ("l3.conv2", 4, 20, 256, 256, 3, 1, 1, 0, 5),
    ("l4.conv1a", 4, 20, 1024, 512, 1, 1, 0, 0, 1), ("l4.conv2a", 4, 20, 512, 512, 3, 2, 1, 0, 1),
    ("l4.down", 4, 20, 1024, 2048, 1, 2, 0, 0, 1), ("l4.conv3", 2, 10, 512, 2048, 1, 1, 0, 1, 3),
    ("l4.conv1", 2, 10, 2048, 512, 1, 1, 0, 0, 2), ("l4.conv2", 2, 10, 512, 512, 3, 1, 1, 0, 2),
]


def main():
    B = int(os.environ.get("B", "256"))
    FLAGS = int(os.environ.get("FLAGS", "0"))      # script-level switch, passed to the ABI as an explicit argument
    if os.environ.get("LIB"):                    # A/B another build of the library on the same box
        _lib.LIB_PATH = os.environ["LIB"]
    L = _lib.lib()
    dev = torch.device("cuda:0")
    total = 0.0
    only = os.environ.get("ONLY")
    for name, H, W, Cin, Cout, k, s, pd, res, cnt in LAYERS:
        if only and name not in only.split(","):
            continue
        Ho, Wo = (H + 2 * pd - k) // s + 1, (W + 2 * pd - k) // s + 1
        x = (torch.randn(B, H, W, Cin, device=dev) * 0.5).to(torch.bfloat16)
        w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * k * k) ** -0.5
        ones, zeros = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
        nb = L.i2l_conv_bf16_packed_bytes(Cout, Cin, k, k)
        packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(L.i2l_conv_bn_bf16_pack(w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), ones.data_ptr(),
                                           1e-5, packed.data_ptr(), nb, Cout, Cin, k, k, _lib.stream_ptr()), "pack")
        r = (torch.randn(B, Ho, Wo, Cout, device=dev)).to(torch.bfloat16) if res else None
        y = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
        wsb = L.i2l_conv_bf16_workspace_bytes(B, H, W, Cin, Cout, k, k, s, pd, FLAGS)
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)

        def run():
            _lib.check(L.i2l_conv_bn_act_bf16_fwd(x.data_ptr(), 0, packed.data_ptr(), _lib.ptr(r), y.data_ptr(), B, H, W, Cin, Cout,
                                                  k, k, s, pd, 1, ws.data_ptr(), wsb, FLAGS, _lib.stream_ptr()), "conv")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / n
        M = B * Ho * Wo
        flops = 2.0 * M * Cout * Cin * k * k
        nbytes = 2.0 * (M * Cout * (2 if res else 1) + Cout * Cin * k * k) + 2.0 * B * H * W * Cin / (s * s if k == 1 else 1)
        total += us * cnt
        print(f"{name:10s} M={M:7d} N={Cout:4d} K={Cin*k*k:4d} x{cnt}  {us:7.1f} us  {flops/us/1e6:6.1f} TF/s  {nbytes/us/1e3:6.0f} GB/s")
    print(f"sum over the network (without stem/pools): {total:.0f} us")


if __name__ == "__main__":
    main()
