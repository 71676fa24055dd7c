"""CPU oracle for the ResNet encoder -- TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY UNPINNED: torchvision is not installed and the reference's ResNetEncoder fetches remote
ImageNet weights (encoder.py:185-194), so no output of the reference itself exists for
this path.  This is a functional fp32 restatement of the published torchvision ResNet v1.5 forward
(conv7x7/2 + BN + ReLU + maxpool3x3/2; BasicBlock / Bottleneck stages with the stride on the 3x3;
adaptive average pool) in the child order the reference slices (encoder.py:198-206), followed by
Flatten + Linear + ReLU (encoder.py:242-247), driven by a state_dict with the reference's key names.
"""
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BLOCKS = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]),
          "resnet50": ("bottleneck", [3, 4, 6, 3]), "resnet101": ("bottleneck", [3, 4, 23, 3]),
          "resnet152": ("bottleneck", [3, 8, 36, 3])}


def _bn(sd: SD, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training=False, eps=1e-5)


def resnet_trunk(sd: SD, model_name: str, x: torch.Tensor, prefix: str = "encoder.resnet.") -> torch.Tensor:
    kind, counts = BLOCKS[model_name]
    x = F.relu(_bn(sd, prefix + "1", F.conv2d(x, sd[prefix + "0.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for li, n in enumerate(counts):
        for bi in range(n):
            p = f"{prefix}{4 + li}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = x
            if kind == "bottleneck":
                o = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"])))
                o = F.relu(_bn(sd, p + "bn2", F.conv2d(o, sd[p + "conv2.weight"], stride=stride, padding=1)))
                o = _bn(sd, p + "bn3", F.conv2d(o, sd[p + "conv3.weight"]))
            else:
                o = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1)))
                o = _bn(sd, p + "bn2", F.conv2d(o, sd[p + "conv2.weight"], padding=1))
            if p + "downsample.0.weight" in sd:
                identity = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
            x = F.relu(o + identity)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def resnet_encoder(sd: SD, model_name: str, x: torch.Tensor) -> torch.Tensor:
    feat = resnet_trunk(sd, model_name, x)
    return F.relu(F.linear(feat, sd["encoder.embedding_layer.weight"], sd["encoder.embedding_layer.bias"]))
