"""3x3 stride-1 bf16 conv: the LDS-patch kernel against the implicit-GEMM ring kernel (I2L_FLAG_RESNET_NO_PATCH) and an
fp32 torch convolution of the same bf16 operands, on shapes around every edge of the tiling."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib  # noqa: E402

# (B, H, W, Cin, Cout, residual)
CASES = [
    (8, 8, 40, 128, 128, 0), (8, 4, 20, 256, 256, 0), (12, 2, 10, 512, 512, 0), (4, 16, 80, 64, 64, 0),
    (3, 16, 80, 64, 64, 1), (5, 3, 7, 64, 128, 1), (2, 1, 1, 32, 64, 0), (7, 5, 33, 96, 192, 0), (1, 16, 200, 64, 64, 0),
    (2, 8, 100, 128, 128, 0), (3, 4, 50, 256, 256, 1), (5, 2, 25, 512, 512, 0), (1, 9, 129, 64, 128, 0), (1, 3, 256, 64, 64, 0),
    (2, 7, 128, 64, 128, 0), (3, 1, 13, 64, 64, 0), (256, 2, 10, 512, 512, 0),
]


def main():
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    worst = 0.0
    for B, H, W, Cin, Cout, res in CASES:
        x = (torch.randn(B, H, W, Cin, device=dev) * 0.5).to(torch.bfloat16)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * (Cin * 9) ** -0.5
        gamma, beta = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
        mean, var = torch.randn(Cout, device=dev) * 0.1, torch.rand(Cout, device=dev) + 0.5
        nb = L.i2l_conv_bf16_packed_bytes(Cout, Cin, 3, 3)
        packed = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(L.i2l_conv_bn_bf16_pack(w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                           1e-5, packed.data_ptr(), nb, Cout, Cin, 3, 3, _lib.stream_ptr()), "pack")
        r = torch.randn(B, H, W, Cout, device=dev).to(torch.bfloat16) if res else None
        outs = []
        variants = [0] + [_lib.flag_resnet_patch_shape(n) for n in range(1, 6)] + [_lib.FLAG_RESNET_NO_PATCH]
        for flags in variants:
            y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
            wsb = L.i2l_conv_bf16_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, 1, flags)
            ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=dev)
            _lib.check(L.i2l_conv_bn_act_bf16_fwd(x.data_ptr(), 0, packed.data_ptr(), _lib.ptr(r), y.data_ptr(), B, H, W, Cin,
                                                  Cout, 3, 3, 1, 1, 1, ws.data_ptr(), wsb, flags, _lib.stream_ptr()), "conv")
            outs.append(y.float())
        torch.cuda.synchronize()
        sc = gamma / torch.sqrt(var + 1e-5)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), padding=1) * sc[None, :, None, None] \
            + (beta - mean * sc)[None, :, None, None]
        if res:
            ref = ref + r.float().permute(0, 3, 1, 2)
        ref = torch.relu(ref).permute(0, 2, 3, 1)
        e_ref = max(((o - ref).abs() / (ref.abs() + 1.0)).max().item() for o in outs[:-1])
        e_ring = max(((o - outs[-1]).abs() / (outs[-1].abs() + 1.0)).max().item() for o in outs[:-1])
        nan = sum(int(torch.isnan(o).sum().item()) for o in outs)
        worst = max(worst, e_ref)
        print(f"B={B:3d} H={H:2d} W={W:3d} Cin={Cin:3d} Cout={Cout:3d} res={res}: vs fp32 conv {e_ref:.2e}  vs ring kernel {e_ring:.2e}  nan {nan}",
              flush=True)
        assert nan == 0 and e_ref < 1.2e-2 and e_ring < 1.2e-2, "mismatch"
    print("worst vs fp32 conv of the same bf16 operands:", worst)


if __name__ == "__main__":
    main()
