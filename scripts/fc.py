import sys, torch, time
sys.path.insert(0, "hmer-img2latex_amd")
from img2latex_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
M, K, N = 256, 40960, 256
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
nb = L.i2l_linear_workspace_bytes(M, K, N)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    return L.i2l_linear_bias_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, K, N, 1, ws.data_ptr(), nb, _lib.stream_ptr())
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("fc us", e0.elapsed_time(e1) * 1000 / 50)
