"""CPU-side checks: the C-ABI library loads and exports every symbol include/*.h declares,
the host classes keep the reference's surface, and the product path refuses CPU tensors."""
import json
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from img2latex_amd import _lib, synth
from img2latex_amd.model import Attention, CNNEncoder, LSTMDecoder, Seq2SeqModel

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "img2latex_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(i2l_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    handle = _lib.lib()                       # binds every symbol, AttributeError if one is missing
    assert handle.i2l_version() >= 100
    assert handle.i2l_error_string(-2).decode().startswith("dimension")
    # size queries are pure host code: callable without a GPU
    assert handle.i2l_decoder_workspace_bytes(256, 512, 256, 256, 1) > 0
    assert handle.i2l_decoder_workspace_bytes(0, 512, 256, 256, 1) == 0
    assert handle.i2l_linear_workspace_bytes(256, 40960, 256) > 0


def test_library_has_no_global_switches():
    """include/img2latex_hip.h promises "no global state": kernel selection is the explicit `flags` argument, the
    sources read no environment variable, and the Python layer never edits os.environ."""
    csrc = os.path.join(REPO, "hmer-img2latex_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    pkg = os.path.join(REPO, "hmer-img2latex_amd", "img2latex_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                assert "os.environ[" not in open(os.path.join(root, f)).read(), f
    header = open(os.path.join(REPO, "include", "img2latex_hip.h")).read()
    for name in ("I2L_FLAG_EXACT_FP32", "I2L_FLAG_NO_GROUP", "I2L_FLAG_RESNET_NO_RING", "I2L_FLAG_RESNET_IM2COL_STEM"):
        assert name in header
    assert (_lib.FLAG_EXACT_FP32, _lib.FLAG_NO_GROUP, _lib.FLAG_RESNET_NO_RING, _lib.FLAG_RESNET_IM2COL_STEM) == (1, 2, 4, 8)
    # the stem's im2col variant needs its image in the workspace; the fused stem needs none (host-side size query)
    L = _lib.lib()
    assert L.i2l_conv_bf16_workspace_bytes(2, 64, 320, 3, 64, 7, 7, 2, 3, 0) == 256
    assert L.i2l_conv_bf16_workspace_bytes(2, 64, 320, 3, 64, 7, 7, 2, 3, _lib.FLAG_RESNET_IM2COL_STEM) > 2 * 32 * 160 * 147 * 2


def test_library_has_no_file_scope_mutable_state():
    """caller-owned resources, explicit stream, no global state.  r03 kept a per-device table of side
    streams behind a mutex; they live in caller-owned i2l_lanes objects now.  Every data / bss
    symbol of the library must be a kernel handle of the HIP fat binary (demangles like a function), a FUNCTION-LOCAL
    static (`f(...)::name`: the per-device "attribute already set" bit masks, an idempotent cache) or toolchain furniture."""
    import re
    import subprocess
    out = subprocess.run(["nm", "-C", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    toolchain = ("DW.ref.", "_DYNAMIC", "_GLOBAL_OFFSET_TABLE_", "__hip_", "__do_", "completed.", "__dso_handle", "__TMC_END__",
                 "typeinfo for", "vtable for", "typeinfo name for", "guard variable for", "__bss_start", "_edata", "_end",
                 "__frame_dummy", "__JCR", "__CTOR", "__DTOR", "__init_array", "__fini_array", "_fini", "_init", "__data_start",
                 "data_start", "__gmon_start__", "__EH_FRAME", "__GNU_EH", "__FRAME_END__", "__cuda", "__hipRegister",
                 "__dummy", "__abi_tag", "__fini", "__init")
    offenders, locals_ = [], []
    for line in out.splitlines():
        parts = line.split(None, 2)
        if len(parts) < 3 or parts[1] not in "bBdD":
            continue
        name = parts[2]
        if name.startswith(toolchain):
            continue
        if re.search(r"\)(?: const)?::[A-Za-z_]\w*$", name) or re.match(r"i2l_\w+::[A-Za-z_]\w*$", name):
            locals_.append(name)           # f(...)::x, or extern "C" i2l_f::x
            continue
        if name.endswith(")"):             # a __global__ function's host-side handle
            continue
        offenders.append(name)
    assert not offenders, offenders
    # the function-local statics are the attribute caches and nothing else
    assert all(n.rsplit("::", 1)[1].startswith(("attr", "attr_done", "attr8")) for n in locals_), locals_
    src = open(os.path.join(REPO, "hmer-img2latex_amd", "csrc", "api.hip")).read()
    assert "std::mutex" not in src and "g_side" not in src


def test_state_dict_keys_match_reference():
    cfg = synth.model_config(lstm_layers=2, attention=True, embedding_dim=32, hidden_dim=64, vocab_size=50,
                             channels=1, img_height=16, img_width=32, conv_filters=(4, 8, 16))
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    ours = list(m.state_dict().keys())
    want = list(synth.make_state_dict(cfg).keys())
    assert sorted(ours) == sorted(want)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg).items()}, strict=True)


def test_default_init_matches_reference_under_same_seed():
    d = np.load(os.path.join(GOLDEN, "init_parity.npz"))
    cfg = json.loads(str(d["cfg_json"]))
    torch.manual_seed(1234)
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    names = json.loads(str(d["names"]))
    sd = m.state_dict()
    assert list(sd.keys()) == names
    got = np.array([synth.checksum(sd[k].numpy()) for k in names])
    np.testing.assert_allclose(got, d["checksums"], rtol=0, atol=0)


def test_constructor_defaults_and_errors():
    enc = CNNEncoder()
    assert (enc.img_height, enc.img_width, enc.channels, enc.embedding_dim) == (64, 800, 1, 256)   # encoder.py:51-64
    assert enc.embedding_layer.in_features == 128 * 8 * 100
    dec = LSTMDecoder(vocab_size=10)
    assert (dec.embedding_dim, dec.hidden_dim, dec.max_seq_length, dec.lstm_layers, dec.dropout) == (256, 256, 141, 1, 0.1)
    assert dec.use_attention and isinstance(dec.attention, Attention)                                # decoder.py:32
    m = Seq2SeqModel()
    assert m.vocab_size == 100 and not m.decoder.use_attention                                       # seq2seq.py:51,90
    assert m.encoder.embedding_layer.in_features == 128 * (50 // 8) * (200 // 8)
    with pytest.raises(ValueError):
        Seq2SeqModel(model_type="transformer")
    with pytest.raises(NotImplementedError):
        CNNEncoder(kernel_size=5)


def test_no_cpu_fallback():
    cfg = synth.model_config(embedding_dim=32, hidden_dim=64, vocab_size=50, channels=1, img_height=16,
                             img_width=32, conv_filters=(4, 8, 16))
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).eval()
    x = torch.zeros(2, 1, 16, 32)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.encoder(x)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.decoder.decode_step(torch.zeros(2, 32), torch.ones(2, 1, dtype=torch.long), None)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m.inference(x, 1, 2, max_length=4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "hmer-img2latex_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "img2latex_oracle" not in src and "import oracle" not in src, f


def test_resnet_encoder_surface():
    """state_dict keys follow the reference's nn.Sequential(*list(backbone.children())[:-1]) slicing
    (encoder.py:198-199): conv1 -> resnet.0, bn1 -> resnet.1, layer1..4 -> resnet.4..7."""
    from img2latex_amd.model import ResNetEncoder
    enc = ResNetEncoder(model_name="resnet50", embedding_dim=128)
    assert (enc.img_height, enc.img_width, enc.channels) == (64, 800, 3)               # encoder.py:163-170
    keys = set(enc.state_dict().keys())
    for k in ("resnet.0.weight", "resnet.1.running_mean", "resnet.1.num_batches_tracked", "resnet.4.0.conv1.weight",
              "resnet.4.0.downsample.0.weight", "resnet.4.0.downsample.1.bias", "resnet.4.2.bn3.weight",
              "resnet.5.0.conv2.weight", "resnet.6.5.conv3.weight", "resnet.7.2.bn3.running_var",
              "embedding_layer.weight", "embedding_layer.bias"):
        assert k in keys, k
    assert enc.state_dict()["resnet.5.0.conv2.weight"].shape == (128, 128, 3, 3)
    assert enc.state_dict()["resnet.5.0.downsample.0.weight"].shape == (512, 256, 1, 1)
    assert enc.embedding_layer.in_features == 2048
    assert sum(p.numel() for n, p in enc.named_parameters() if n.startswith("resnet.")) == 23508032   # resnet50 minus fc
    # freeze_backbone=True: everything frozen except layer4 (encoder.py:201-210)
    assert not enc.resnet[4][0].conv1.weight.requires_grad and enc.resnet[7][0].conv1.weight.requires_grad
    assert ResNetEncoder(model_name="resnet18").embedding_layer.in_features == 512
    with pytest.raises(ValueError):
        ResNetEncoder(model_name="resnet42")
    m = Seq2SeqModel("resnet_lstm", 50, {"model_name": "resnet18", "embedding_dim": 32}, {"hidden_dim": 64})
    assert m.model_type == "resnet_lstm" and any(k.startswith("encoder.resnet.7.") for k in m.state_dict())


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "I2L_DIST_BACKEND")):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus N` with no launcher around it starts N ranks itself and rank 0 prints ONE line with
    n_gpus = N (--launch-probe: the rendezvous only, gloo on the CPU -- the same code path the GPU modes take)."""
    import json
    r = _run_bench(["--gpus", "3", "--launch-probe"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["ranks_seen"] == 6.0          # ranks 0, 1, 2 all took part
    # the driver's largest case, through the train mode's argument path: 8 ranks rendezvous and all take part
    r = _run_bench(["--gpus", "8", "--mode", "train", "--launch-probe"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 36.0


def test_bench_refuses_a_mismatched_world():
    """Never a line whose n_gpus is not what was asked for: a launcher-provided WORLD_SIZE that differs from --gpus
    is refused (also WORLD_SIZE=1 with --gpus 2, which r02 silently ran as one rank), and without enough visible GPUs
    the self-launch refuses as well instead of running a smaller job."""
    r = _run_bench(["--gpus", "2", "--launch-probe"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    import torch
    if torch.cuda.device_count() < 2:
        r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
        assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
