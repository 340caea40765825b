"""Wave.from_file / Wave.save host logic with a stand-in codec: layout,
metadata, frame windows, the encoding -> libsndfile subtype table of the reference
(src/torchfx/wave.py:538-560) and directory creation."""
import sys

import numpy as np
import pytest
import torch

import torchfx_amd as fx
from tests import _fake_soundfile as sf


@pytest.fixture
def soundfile(monkeypatch):
    monkeypatch.setitem(sys.modules, "soundfile", sf)
    sf.written.clear()
    return sf


def test_from_file_cpu_layout_metadata_and_window(tmp_path, soundfile):
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=(1000, 3), dtype=np.int16)
    p = tmp_path / "a.wav"
    sf.make(p, pcm, 44100)
    w = fx.Wave.from_file(p)
    assert w.fs == 44100 and w.ys.shape == (3, 1000) and w.ys.dtype == torch.float32 and w.ys.is_contiguous()
    assert np.array_equal(w.ys.numpy(), (pcm.astype(np.float32) / np.float32(32768)).T)
    assert w.metadata == {"num_frames": 1000, "num_channels": 3, "subtype": "PCM_16", "format": "WAV"}
    part = fx.Wave.from_file(p, frame_offset=100, num_frames=50)
    assert part.ys.shape == (3, 50) and np.array_equal(part.ys.numpy(), w.ys.numpy()[:, 100:150])
    assert fx.Wave.from_file(str(p), frame_offset=990).ys.shape == (3, 10)


@pytest.mark.parametrize("kw,subtype", [
    (dict(), None), (dict(bits_per_sample=24), "PCM_24"), (dict(encoding="PCM_F"), "FLOAT"),
    (dict(encoding="PCM_S", bits_per_sample=16), "PCM_16"), (dict(encoding="PCM_U", bits_per_sample=8), "PCM_U8"),
    (dict(encoding="PCM_U", bits_per_sample=16), "PCM_16"), (dict(encoding="PCM_F", bits_per_sample=32), "FLOAT"),
    (dict(encoding="PCM_F", bits_per_sample=64), "DOUBLE"), (dict(encoding="ULAW", bits_per_sample=8), "ULAW8"),
    (dict(encoding="PCM_S"), None)])
def test_save_subtype_table_and_layout(tmp_path, soundfile, kw, subtype):
    ys = torch.arange(12, dtype=torch.float32).reshape(2, 6) / 16
    target = tmp_path / "deep" / "er" / "out.flac"
    fx.Wave(ys, 8000).save(target, **kw)
    rec = sf.written[-1]
    assert target.parent.is_dir()
    assert rec["format"] == "FLAC" and rec["subtype"] == subtype and rec["fs"] == 8000
    assert rec["shape"] == (6, 2) and np.array_equal(rec["data"], ys.numpy().T)
    fx.Wave(ys, 8000).save(tmp_path / "x.unknown")
    assert sf.written[-1]["format"] == "WAV"
    fx.Wave(ys, 8000).save(tmp_path / "y.ogg", format="WAV")
    assert sf.written[-1]["format"] == "WAV"
