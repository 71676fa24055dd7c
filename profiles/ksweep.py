import os, sys, torch
sys.path.insert(0, "/root/repo/hmer-img2latex_amd")
sys.path.insert(0, os.path.join(os.getcwd(), "hmer-img2latex_amd"))
from img2latex_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def run_case(B, H, W, Cin, Cout, k, s, pd, res=0, n=30):
    Ho, Wo = (H + 2 * pd - k) // s + 1, (W + 2 * pd - k) // s + 1
    x = (torch.randn(B, H, W, Cin, device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * k * k) ** -0.5
    ones, zeros = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    nb = L.i2l_conv_bf16_packed_bytes(Cout, Cin, k, k)
    packed = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(L.i2l_conv_bn_bf16_pack(w.data_ptr(), ones.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), ones.data_ptr(), 1e-5, packed.data_ptr(), nb, Cout, Cin, k, k, _lib.stream_ptr()), "pack")
    y = torch.empty(B, Ho, Wo, Cout, dtype=torch.bfloat16, device=dev)
    wsb = L.i2l_conv_bf16_workspace_bytes(B, H, W, Cin, Cout, k, k, s, pd, 0)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
    def run():
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(x.data_ptr(), 0, packed.data_ptr(), None, y.data_ptr(), B, H, W, Cin, Cout, k, k, s, pd, 1, ws.data_ptr(), wsb, 0, _lib.stream_ptr()), "conv")
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for (B, H, W) in [(256, 4, 20), (64, 4, 20), (16, 4, 20)]:
    for Cout in (256,):
        for Cin in (64, 128, 256, 512, 1024, 2048, 4096):
            us = run_case(B, H, W, Cin, Cout, 1, 1, 0)
            M = B * H * W
            print(f"M={M} N={Cout} K={Cin} KT={Cin//64} tiles={(M//128)*(Cout//128)} {us:.1f} us", flush=True)
