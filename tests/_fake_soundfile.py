"""TEST-ONLY stand-in for the ``soundfile`` package (absent from the image): "audio files" are .npz
archives holding int16 or float32 frames.  Only the calls Wave.from_file / Wave.save make."""
import types

import numpy as np

written = []


def _load(path):
    z = np.load(path, allow_pickle=False)
    return z["frames"], int(z["fs"]), str(z["subtype"]), str(z["format"])


def make(path, frames, fs, subtype="PCM_16", fmt="WAV"):
    with open(path, "wb") as fh:
        np.savez(fh, frames=frames, fs=fs, subtype=subtype, format=fmt)


def info(path):
    frames, fs, subtype, fmt = _load(path)
    return types.SimpleNamespace(frames=frames.shape[0], channels=frames.shape[1], subtype=subtype, format=fmt,
                                 samplerate=fs)


def read(path, start=0, stop=None, dtype="float64", always_2d=False):
    frames, fs, subtype, _ = _load(path)
    sel = frames[start:stop]
    if dtype == "int16":
        assert frames.dtype == np.int16
        return sel.copy(), fs
    if frames.dtype == np.int16:                       # libsndfile: PCM16 read as float = v / 32768
        sel = sel.astype(np.float32) / np.float32(32768.0)
    return sel.astype(dtype), fs


def write(path, data, samplerate, format=None, subtype=None):  # noqa: A002
    written.append(dict(path=str(path), shape=tuple(data.shape), fs=samplerate, format=format, subtype=subtype,
                        data=np.array(data, copy=True)))


class SoundFile:
    """Write-mode context manager: collects the blocks written to it (``StreamProcessor.process_file``)."""

    def __init__(self, path, mode="w", samplerate=None, channels=None, format=None, subtype=None):  # noqa: A002
        assert mode == "w"
        self.rec = dict(path=str(path), fs=samplerate, channels=channels, format=format, subtype=subtype, blocks=[])

    def __enter__(self):
        return self

    def write(self, data):
        data = np.asarray(data)
        assert data.ndim == 2 and data.shape[1] == self.rec["channels"], data.shape
        self.rec["blocks"].append(np.array(data, copy=True))

    def __exit__(self, *exc):
        blocks = self.rec.pop("blocks")
        self.rec["data"] = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, self.rec["channels"]), np.float32)
        self.rec["shape"] = self.rec["data"].shape
        written.append(self.rec)
        return False
