"""Time i2l_bottleneck_join_bf16_fwd against the two launches it replaces at layer1's size (B=256, 16x80 positions)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
B, H, W = 256, 16, 80
for n2 in (64, 128):
    o2 = (torch.randn(B, H, W, 64, device=dev) * 0.7).to(torch.bfloat16)
    ident = torch.randn(B, H, W, 256, device=dev).to(torch.bfloat16)

    def packed(cout, cin):
        w = torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5
        one, zero = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        nb = L.i2l_conv_bf16_packed_bytes(cout, cin, 1, 1)
        buf = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(L.i2l_conv_bn_bf16_pack(w.data_ptr(), one.data_ptr(), zero.data_ptr(), zero.data_ptr(), one.data_ptr(), 1e-5,
                                           buf.data_ptr(), nb, cout, cin, 1, 1, _lib.stream_ptr()), "pack")
        return buf
    p3, p1 = packed(256, 64), packed(n2, 256)
    ws = torch.empty(4096, dtype=torch.uint8, device=dev)
    y = torch.empty(B, H, W, 256, dtype=torch.bfloat16, device=dev)
    z = torch.empty(B, H, W, n2, dtype=torch.bfloat16, device=dev)

    def two():
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(o2.data_ptr(), 0, p3.data_ptr(), ident.data_ptr(), y.data_ptr(), B, H, W, 64, 256,
                                              1, 1, 1, 0, 1, ws.data_ptr(), 4096, 0, _lib.stream_ptr()), "conv3")
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(y.data_ptr(), 0, p1.data_ptr(), None, z.data_ptr(), B, H, W, 256, n2,
                                              1, 1, 1, 0, 1, ws.data_ptr(), 4096, 0, _lib.stream_ptr()), "conv1")

    def one():
        _lib.check(L.i2l_bottleneck_join_bf16_fwd(o2.data_ptr(), p3.data_ptr(), ident.data_ptr(), y.data_ptr(), p1.data_ptr(),
                                                  z.data_ptr(), B * H * W, 64, 256, n2, _lib.stream_ptr()), "join")
    for name, fn in (("two launches", two), ("join", one), ("two launches", two), ("join", one)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"n2={n2:3d} {name:13s} {e0.elapsed_time(e1) / 30 * 1000:7.1f} us", flush=True)
