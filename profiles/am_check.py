import sys, torch
sys.path.insert(0, "hmer-img2latex_amd")
from img2latex_amd import _lib
L = _lib.lib()
torch.manual_seed(0)
for (B, Cin, H, W, Cout) in [(4, 3, 64, 320, 32), (4, 32, 32, 160, 64), (4, 64, 16, 80, 128), (2, 16, 10, 14, 64), (3, 3, 18, 34, 32)]:
    x = torch.randn(B, Cin, H, W, device="cuda"); w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)); b = torch.randn(Cout, device="cuda")
    outs = []
    for flags in (0, 1):
        y = torch.empty(B, Cout, H // 2, W // 2, device="cuda"); am = torch.full(y.shape, 9, dtype=torch.uint8, device="cuda")
        nb = L.i2l_conv_workspace_bytes(Cin, Cout); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
        assert L.i2l_conv3x3_relu_pool2_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), am.data_ptr(), B, Cin, H, W, Cout, ws.data_ptr(), nb, flags, _lib.stream_ptr()) == 0
        outs.append((y, am))
    (y0, a0), (y1, a1) = outs
    pos = y1 > 0
    print((B, Cin, H, W, Cout), "y max diff", float((y0 - y1).abs().max()), "argmax mismatch (all)", int((a0 != a1).sum()), "of", a0.numel(),
          "mismatch where y>0", int(((a0 != a1) & pos).sum()), "max am", int(a0.max()))
