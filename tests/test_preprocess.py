"""Image preprocessing (reference data/utils.py:18-110, data/transforms.py:26-56; Pillow 12.2 LANCZOS).

CPU: the oracle restatement and the library's host coefficient helper against the reference's golden outputs.
GPU: the HIP kernels, bit-identical float32 tensors for the whole ragged batch."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN
import preprocess_oracle as PO
from img2latex_amd import _lib, synth


def make_image(seed, h, w, c):          # the generator of tests/golden/make_golden_preprocess.py
    base = synth.uniform(seed, "img", (h, w, c), 0.0, 1.0)
    strokes = (synth.uniform(seed + 1, "mask", (h, w, 1), 0.0, 1.0) < 0.18)
    img = np.where(strokes, base * 90.0, 200.0 + base * 55.0)
    return np.clip(np.round(img), 0, 255).astype(np.uint8).reshape((h, w) if c == 1 else (h, w, 3))


def cases():
    d = np.load(f"{GOLDEN}/preprocess.npz")
    return d, [tuple(r) for r in d["cases"].tolist()]


def test_oracle_bit_exact_vs_reference():
    d, cs = cases()
    for i, (h, w, c, th, tw, oc) in enumerate(cs):
        got = PO.load_image_from_array(make_image(1000 + 10 * i, h, w, c), (th, tw), oc, True)
        assert got.dtype == np.float32 and got.shape == d[f"out{i}"].shape
        assert np.array_equal(got, d[f"out{i}"]), i
    for i in (0, 5):
        h, w, c, th, tw, oc = cs[i]
        assert np.array_equal(PO.load_image_from_array(make_image(1000 + 10 * i, h, w, c), (th, tw), oc, False), d[f"raw{i}"])


def test_host_coefficients_equal_oracle():
    """i2l_lanczos_coeffs (C, libm) == the oracle's restatement of Pillow's precompute_coeffs (Python, libm)."""
    L = _lib.lib()
    for (a, b) in [(150, 240), (500, 160), (2400, 768), (64, 64), (40, 64), (200, 64), (23, 60), (17, 32), (300, 14)]:
        ks, bo, ko = PO.precompute_coeffs(a, 0.0, float(a), b)
        assert L.i2l_lanczos_ksize(a, b) == ks
        bounds = np.zeros((b, 2), np.int32)
        kk = np.zeros((b, ks), np.int32)
        assert L.i2l_lanczos_coeffs(a, b, bounds.ctypes.data, kk.ctypes.data) == 0
        assert np.array_equal(bounds, bo) and np.array_equal(kk, ko), (a, b)


def test_batched_coefficient_helper_equals_single_calls():
    """i2l_resample_coeffs_batch (host threads inside the call) writes, per entry, exactly what i2l_resample_coeffs
    writes: bounds (out, 2) then weights (out, ksize) at the entry's offset."""
    L = _lib.lib()
    pairs = [(150, 240), (500, 160), (2400, 768), (64, 64), (40, 64), (200, 64), (23, 60), (17, 32), (300, 14), (811, 320)]
    for flt in (_lib.FILTER_LANCZOS, _lib.FILTER_BICUBIC):
        ks = [L.i2l_resample_ksize(flt, a, b) for a, b in pairs]
        sizes = [b * (2 + k) for (a, b), k in zip(pairs, ks)]
        offs = np.zeros(len(pairs), np.int64)
        offs[1:] = np.cumsum(sizes[:-1])
        out = np.full(sum(sizes), -7, np.int32)
        ins, outs = np.array([a for a, _ in pairs], np.int32), np.array([b for _, b in pairs], np.int32)
        for threads in (1, 4, 64):
            out[:] = -7
            assert L.i2l_resample_coeffs_batch(flt, len(pairs), ins.ctypes.data, outs.ctypes.data, offs.ctypes.data,
                                               out.ctypes.data, threads) == 0
            for (a, b), k, o in zip(pairs, ks, offs.tolist()):
                bounds, kk = np.zeros((b, 2), np.int32), np.zeros((b, k), np.int32)
                assert L.i2l_resample_coeffs(flt, a, b, bounds.ctypes.data, kk.ctypes.data) == 0
                assert np.array_equal(out[o:o + 2 * b].reshape(b, 2), bounds), (a, b, threads)
                assert np.array_equal(out[o + 2 * b:o + b * (2 + k)].reshape(b, k), kk), (a, b, threads)
    assert L.i2l_resample_coeffs_batch(9, 1, ins.ctypes.data, outs.ctypes.data, offs.ctypes.data, out.ctypes.data, 1) < 0


@pytest.mark.gpu
def test_device_preprocessing_bit_exact():
    from img2latex_amd.data import preprocess_batch
    d, cs = cases()
    groups = {}
    for i, (h, w, c, th, tw, oc) in enumerate(cs):
        groups.setdefault((th, tw, oc), []).append(i)
    for (th, tw, oc), idx in groups.items():                       # one ragged batch per output configuration
        imgs = [make_image(1000 + 10 * i, *cs[i][:3]) for i in idx]
        out = preprocess_batch(imgs, (th, tw), oc, True).cpu().numpy()
        for j, i in enumerate(idx):
            assert np.array_equal(out[j], d[f"out{i}"]), (i, cs[i])
    raw = preprocess_batch([make_image(1000, *cs[0][:3])], cs[0][3:5], cs[0][5], False).cpu().numpy()[0]
    assert np.array_equal(raw, d["raw0"])


@pytest.mark.gpu
def test_device_preprocessing_full_batch_vs_oracle():
    """256 page-like images of assorted sizes -> (256, 3, 64, 320): every pixel equal to the oracle's."""
    from img2latex_amd.data import preprocess_batch
    sizes = [(30 + (7 * k) % 90, 80 + (53 * k) % 700, 1 + 2 * (k % 2)) for k in range(256)]
    imgs = [make_image(5000 + k, h, w, c) for k, (h, w, c) in enumerate(sizes)]
    out = preprocess_batch(imgs, (64, 320), 3, True)
    assert out.shape == (256, 3, 64, 320) and out.is_cuda
    got = out.cpu().numpy()
    for k in range(0, 256, 17):
        assert np.array_equal(got[k], PO.load_image_from_array(imgs[k], (64, 320), 3, True)), k


def test_vectorised_plan_arithmetic_and_host_packing():
    """The host side of the r04 preprocessing path, without a GPU: the numpy restatement of the table sizes and of the
    vertical extent (data/preprocess.py: _ksizes, _extents) equals the library's own numbers for random sizes and both
    filters; i2l_pack_host gathers ragged buffers exactly, for 1 .. 64 threads."""
    from img2latex_amd.data.preprocess import _extents, _ksizes
    L = _lib.lib()
    rng = np.random.default_rng(3)
    ins = np.concatenate([rng.integers(1, 3000, 300), [1, 2, 64, 64, 4000]]).astype(np.int64)
    outs = np.concatenate([rng.integers(1, 2000, 300), [1, 7, 64, 65, 50]]).astype(np.int64)
    for flt in (_lib.FILTER_LANCZOS, _lib.FILTER_BICUBIC):
        ks = _ksizes(flt, ins, outs)
        first, last = _extents(flt, ins, outs)
        for a, b, k, f0, l0 in zip(ins.tolist(), outs.tolist(), ks.tolist(), first.tolist(), last.tolist()):
            assert L.i2l_resample_ksize(flt, a, b) == k, (a, b)
            if k > 512 or a * b > 400000:
                continue
            bounds, kk = np.zeros((b, 2), np.int32), np.zeros((b, k), np.int32)
            assert L.i2l_resample_coeffs(flt, a, b, bounds.ctypes.data, kk.ctypes.data) == 0
            assert (int(bounds[0, 0]), int(bounds[-1, 0] + bounds[-1, 1])) == (f0, l0), (a, b)
    bufs = [rng.integers(0, 255, int(n), dtype=np.uint8) for n in rng.integers(0, 200000, 37)]
    sizes = np.array([b.size for b in bufs], np.int64)
    offs = np.zeros(len(bufs), np.int64)
    offs[1:] = np.cumsum(sizes[:-1])
    ptrs = np.array([b.__array_interface__["data"][0] for b in bufs], np.uint64)
    want = np.concatenate(bufs)
    for threads in (1, 3, 8, 64):
        dst = np.full(int(sizes.sum()) + 5, 77, np.uint8)
        assert L.i2l_pack_host(ptrs.ctypes.data, sizes.ctypes.data, offs.ctypes.data, len(bufs), dst.ctypes.data, threads) == 0
        assert np.array_equal(dst[:-5], want) and bool((dst[-5:] == 77).all()), threads
    assert L.i2l_pack_host(None, sizes.ctypes.data, offs.ctypes.data, 3, want.ctypes.data, 1) < 0


@pytest.mark.gpu
def test_device_built_tables_equal_the_host_tables():
    """i2l_resample_coeffs_device (r04: the batch's resampling tables built on the device from its size list) against
    i2l_resample_coeffs (host, libm = Pillow's arithmetic) for ~700 (source, target) sizes and both filters: every bound and
    every 22-bit weight.  The device's sin() is not libm's, so a weight MAY differ by one unit where the normalised value
    sits within ~1e-16 of a rounding boundary (odds ~1e-9 per weight): equality is asserted, and a ragged batch goes through
    preprocess_batch with device-built and with host-built tables bit-identically."""
    from img2latex_amd.data import preprocess_batch
    from img2latex_amd.data.preprocess import _ksizes
    L = _lib.lib()
    rng = np.random.default_rng(11)
    ins = np.concatenate([rng.integers(8, 1200, 340), [64, 40, 200, 23, 17, 300, 150, 500, 2400, 811]]).astype(np.int32)
    outs = np.concatenate([rng.integers(8, 1700, 340), [64, 64, 64, 60, 32, 14, 240, 160, 768, 320]]).astype(np.int32)
    for flt in (_lib.FILTER_LANCZOS, _lib.FILTER_BICUBIC):
        ks = _ksizes(flt, ins, outs)
        sizes = outs.astype(np.int64) * (2 + ks)
        offs = np.zeros(len(ins), np.int64)
        offs[1:] = np.cumsum(sizes[:-1])
        want = np.zeros(int(sizes.sum()), np.int32)
        assert L.i2l_resample_coeffs_batch(flt, len(ins), ins.ctypes.data, outs.ctypes.data, offs.ctypes.data, want.ctypes.data, 16) == 0
        d_in, d_out, d_off = (torch.from_numpy(a).cuda() for a in (ins, outs, offs))
        got = torch.full((want.size,), -9, dtype=torch.int32, device="cuda")
        assert L.i2l_resample_coeffs_device(flt, len(ins), d_in.data_ptr(), d_out.data_ptr(), d_off.data_ptr(), got.data_ptr(),
                                            int(outs.max()), _lib.stream_ptr()) == 0
        got = got.cpu().numpy()
        diff = np.nonzero(got != want)[0]
        assert diff.size == 0, (flt, diff[:5], got[diff[:5]], want[diff[:5]])
    sizes = [(30 + (7 * k) % 90, 80 + (53 * k) % 700, 1 + 2 * (k % 2)) for k in range(96)]
    imgs = [make_image(7000 + k, h, w, c) for k, (h, w, c) in enumerate(sizes)]
    a = preprocess_batch(imgs, (64, 320), 3, True, tables="device")
    b = preprocess_batch(imgs, (64, 320), 3, True, tables="host")
    assert torch.equal(a, b)
    c = preprocess_batch(imgs[:9], (64, 800), 1, "symmetric", keep_aspect=False, resample="bicubic", tables="device")
    d = preprocess_batch(imgs[:9], (64, 800), 1, "symmetric", keep_aspect=False, resample="bicubic", tables="host")
    assert torch.equal(c, d)
