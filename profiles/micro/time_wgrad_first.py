"""First conv block's weight gradient at the training shape (64 x 3 x 64 x 320 -> 32): sparse kernel against the GEMM path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib
DEV = torch.device("cuda:0")
L = _lib.lib()
B, Cin, H, W, Cout = 64, 3, 64, 320, 32
torch.manual_seed(0)
x = torch.randn(B, Cin, H, W, device=DEV); w = torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.2; b = torch.randn(Cout, device=DEV)
y = torch.empty(B, Cout, H // 2, W // 2, device=DEV); am = torch.empty(y.shape, dtype=torch.uint8, device=DEV)
nb = L.i2l_conv_workspace_bytes(Cin, Cout); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
assert L.i2l_conv3x3_relu_pool2_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), am.data_ptr(), B, Cin, H, W, Cout, ws.data_ptr(), nb, 0, _lib.stream_ptr()) == 0
dy = torch.randn_like(y); dw = torch.empty_like(w); db = torch.empty_like(b)
nb2 = L.i2l_conv_bwd_workspace_bytes(B, Cin, H, W, Cout); ws2 = torch.empty(nb2, dtype=torch.uint8, device=DEV)
for fl, name in ((0, "sparse kernel"), (_lib.FLAG_CONV_NO_SPARSE_WGRAD, "implicit-im2col GEMM")):
    def run():
        assert L.i2l_conv3x3_relu_pool2_bwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), am.data_ptr(), dy.data_ptr(), None, dw.data_ptr(), db.data_ptr(), B, Cin, H, W, Cout, ws2.data_ptr(), nb2, fl, None, _lib.stream_ptr()) == 0
    for _ in range(5): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): run()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call (dw + db)")
