"""GPU parity -- file / layout edge (8f rank 4).
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,F", [(1, 1), (1, 100003), (2, 44100), (3, 7777), (6, 4096), (64, 5000), (100, 333)])
def test_deinterleave_and_interleave_kernels(C, F):
    e = ext()
    rng = np.random.default_rng(C * 17 + F)
    fr = rng.standard_normal((F, C)).astype(np.float32)
    pl = e.deinterleave_forward(dev(fr))
    assert pl.shape == (C, F) and np.array_equal(pl.cpu().numpy(), fr.T)
    back = e.interleave_forward(pl)
    assert back.shape == (F, C) and np.array_equal(back.cpu().numpy(), fr)
    pcm = rng.integers(-32768, 32767, size=(F, C), dtype=np.int16)
    got = e.deinterleave_forward(dev(pcm)).cpu().numpy()
    assert np.array_equal(got, (pcm.astype(np.float32) / np.float32(32768)).T)      # exact: power-of-two scale
    if F > 10:                                      # a chunk written into the middle of a longer tensor
        out = torch.full((C, F + 20), -7.0, device=DEV)
        e.deinterleave_forward(dev(fr), out, frame_base=13)
        o = out.cpu().numpy()
        assert np.array_equal(o[:, 13:13 + F], fr.T) and (o[:, :13] == -7).all() and (o[:, 13 + F:] == -7).all()
        assert np.array_equal(e.interleave_forward(out, 13, F).cpu().numpy(), fr)


def test_chunked_upload_download_and_file_round_trip(tmp_path, monkeypatch):
    import sys

    import torchfx_amd as fx
    from tests import _fake_soundfile as sf
    from torchfx_amd import io as tio
    monkeypatch.setitem(sys.modules, "soundfile", sf)
    rng = np.random.default_rng(3)
    fr = rng.standard_normal((50_001, 5)).astype(np.float32)
    for chunk in (1 << 22, 7000, 50_001, 1):
        if chunk == 1 and fr.shape[0] > 2000:
            small = fr[:1500]
            assert np.array_equal(tio.upload_interleaved(small, DEV, 1).cpu().numpy(), small.T)
            continue
        up = tio.upload_interleaved(fr, DEV, chunk)
        assert np.array_equal(up.cpu().numpy(), fr.T), chunk
        assert np.array_equal(tio.download_interleaved(up, chunk), fr), chunk
    pcm = rng.integers(-32768, 32767, size=(30_000, 2), dtype=np.int16)
    p = tmp_path / "song.wav"
    sf.make(p, pcm, 48000)
    ref = (pcm.astype(np.float32) / np.float32(32768)).T
    for on_dev in (False, True):
        w = fx.Wave.from_file(p, device=DEV, pcm16_on_device=on_dev)
        assert w.ys.is_cuda and w.fs == 48000 and w.metadata["subtype"] == "PCM_16"
        assert np.array_equal(w.ys.cpu().numpy(), ref), on_dev
    w.save(tmp_path / "out" / "copy.wav", encoding="PCM_S", bits_per_sample=16)
    rec = sf.written[-1]
    assert rec["subtype"] == "PCM_16" and np.array_equal(rec["data"], ref.T)
