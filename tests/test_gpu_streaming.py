"""GPU parity -- streaming / realtime callers (8f rank 1): chunked == one shot, HIP-graph replay, file streaming.
Tolerances and helpers: tests/gpu_common.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.gpu_common import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_streaming_chunks_equal_one_shot_on_device():
    from scipy.signal import firwin
    from torchfx_amd import filter as F
    from torchfx_amd.realtime import StatefulFIR, StreamProcessor
    x = rnd((4, 300000), 21)
    taps = firwin(1024, 5000, fs=48000)

    def effects():
        return [F.LoButterworth(2000, order=6), F.ParametricEQ(1000, 2.0, 3.0), StatefulFIR(taps)]
    whole = dev(x)
    for e in effects():
        e.fs = 48000
        whole = e(whole)
    for chunk in (65536, 4096, 1000):
        out = StreamProcessor(effects(), chunk_size=chunk, device=DEV).process_tensor(torch.from_numpy(x), 48000)
        close(out, whole.cpu().numpy(), 2e-6, f"chunk={chunk}")
    two = effects()[:2]
    for e in two:
        e.fs = 48000
        e.compute_coefficients()
    ref = O.chain_forward(x, np.vstack([e._sos.numpy() for e in two]), [O.flipped_kernel(taps)])
    close(whole, ref, TOL_CONV_F32, "one-shot vs oracle")


@pytest.mark.parametrize("C,T,K", [(2, 4096, 1024), (16, 1 << 21, 65536)])
def test_pipeline_is_hip_graph_capturable(C, T, K):
    """No host sync, no allocation and no plan work after warm-up on a stream: a whole step (IIR cascade ->
    overlap-save on the internal two-stream fork/join -> gain+clamp -> per-channel normalise) can be
    captured into a HIP graph and replayed on new input with bit-identical results."""
    from scipy.signal import butter
    e = ext()
    sos = torch.from_numpy(butter(6, 2000 / 24000, output="sos"))
    kf = (np.random.default_rng(K).standard_normal(K) / np.sqrt(K)).astype(np.float32)

    def step(inp):
        y, _, _ = e.sos_forward(inp, None, sos, None, None)
        y = e.fft_conv_forward(y, kf, (K - 1, 0))
        y = e.gain_forward(y, 1.5, True)
        return e.normalize_forward(y, 0.9, e.STAT_ABSMAX, True)

    static_x = dev(rnd((C, T), 77))
    for _ in range(2):
        step(static_x)                                   # warm-up: plans, tables, workspaces
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(static_x)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):           # workspaces are per stream: capture where it warmed up
        out = step(static_x)
    for seed in (78, 79):
        x2 = dev(rnd((C, T), seed))
        static_x.copy_(x2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, step(x2)), seed


@pytest.mark.parametrize("chunk,overlap", [(4096, 0), (8192, 256)])
def test_stream_processor_graph_replay_equals_eager(chunk, overlap):
    """use_graph=True: the per-chunk step (IIR cascade with carried state, stateful FIR history, a
    `+` combination, gain) is captured once and replayed; the output must equal the eager stream
    processor bit for bit, including the ragged last chunk."""
    from torchfx_amd import effect as E
    from torchfx_amd import filter as F
    from torchfx_amd.realtime import StatefulFIR, StreamProcessor

    def make():
        taps = (np.random.default_rng(4).standard_normal(301) / 30).tolist()
        return [F.LoButterworth(3000, order=4, fs=48000), F.ParametricEQ(frequency=800, q=1.0, gain=-3.0, fs=48000),
                StatefulFIR(taps, "fft"),
                F.HiButterworth(100, order=2, fs=48000) + F.BiquadLPF(cutoff=5000, q=0.7, fs=48000),
                E.Gain(0.8, clamp=True)]

    x = dev(rnd((2, chunk * 9 + 1234), 21))
    eager = StreamProcessor(make(), chunk_size=chunk, overlap=overlap, device=DEV).process_tensor(x, 48000)
    sp = StreamProcessor(make(), chunk_size=chunk, overlap=overlap, device=DEV, use_graph=True)
    got = sp.process_tensor(x, 48000)
    assert sp._graph is not None
    assert got.shape == eager.shape and torch.equal(got, eager)
    again = sp.process_tensor(x, 48000)                      # the states went on from the first pass
    eager2 = StreamProcessor(make(), chunk_size=chunk, overlap=overlap, device=DEV)
    eager2.process_tensor(x, 48000)
    assert torch.equal(again, eager2.process_tensor(x, 48000))


@pytest.mark.parametrize("use_graph", [False, True])
def test_realtime_processor_callback_on_device(use_graph):
    """RealtimeProcessor (realtime/processor.py:253-292): host blocks from a backend callback go through pinned staging
    to the device, through the chain (eager, or one replayed HIP graph per block) and back; consecutive callbacks
    are one continuous signal (== the oracle on the whole signal), a staged parameter lands at the next boundary."""
    from scipy.signal import firwin
    from torchfx_amd import effect as E
    from torchfx_amd import filter as F
    from torchfx_amd.realtime import RealtimeProcessor, StatefulFIR, StreamConfig

    class Backend:
        def open_stream(self, config, callback=None):
            self.config, self.callback = config, callback

        def start(self): pass

        def stop(self): pass

        def close(self): pass

        def fire(self, block):
            out = torch.zeros(self.config.channels_out, block.shape[-1])
            self.callback(block, out, block.shape[-1])
            return out

    B, nblocks = 512, 24
    taps = firwin(257, 0.25).astype(np.float32)
    lpf, fir, gain = F.LoButterworth(2000, order=4), StatefulFIR(taps.tolist(), "fft"), E.Gain(0.5)
    be = Backend()
    cfg = StreamConfig(sample_rate=48000, buffer_size=B, channels_in=2, channels_out=2)
    x = rnd((2, B * nblocks + 200), 33)                      # the last block is ragged
    xt = torch.from_numpy(x)
    with RealtimeProcessor([lpf, fir, gain], be, cfg, device=DEV, use_graph=use_graph) as p:
        outs = [be.fire(xt[:, i:i + B]) for i in range(0, B * 12, B)]
        p.set_parameter("2.gain", 2.0)                        # lands at the next buffer boundary
        outs += [be.fire(xt[:, i:i + B]) for i in range(B * 12, x.shape[-1], B)]
        if use_graph:                                             # a chain that is ONE fused launch has no use for a graph
            assert p._runner._graph is not None or p._runner._fused(torch.zeros(xt.shape[0], B, device=DEV))
    y = torch.cat(outs, dim=-1).numpy()
    sos = lpf._sos.cpu().numpy()
    e, _, _ = O.iir_module_forward(x, sos)                    # float32 in, float32 out
    e = O.fir_direct(e.astype(np.float64), O.flipped_kernel(taps).astype(np.float64))
    e[:, :B * 12] *= 0.5
    e[:, B * 12:] *= 2.0
    close(y, e.astype(np.float32), 2e-6, "callback stream")
    # device tensors are taken as they are (no staging), mono goes to every output channel
    be2 = Backend()
    with RealtimeProcessor([E.Gain(2.0)], be2, StreamConfig(48000, 256, channels_in=1, channels_out=2), device=DEV):
        m = dev(rnd((1, 256), 5))
        out = torch.zeros(2, 256, device=DEV)
        be2.callback(m, out, 256)
        assert torch.equal(out[0], m[0] * 2.0) and torch.equal(out[1], m[0] * 2.0)


def test_stream_processor_process_file(tmp_path, monkeypatch):
    """StreamProcessor.process_file / process_file_chunks (src/torchfx/realtime/stream.py:164-347) over the
    stand-in codec: chunked IIR + stateful FIR over a file == the same effects on the whole signal."""
    import sys

    from tests import _fake_soundfile as sf
    from torchfx_amd import filter as F
    from torchfx_amd.realtime import StatefulFIR, StreamProcessor
    monkeypatch.setitem(sys.modules, "soundfile", sf)
    rng = np.random.default_rng(11)
    frames = (rng.standard_normal((50_000, 2)) * 0.3).astype(np.float32)
    src = tmp_path / "in.wav"
    sf.make(src, frames, 44100, subtype="FLOAT")
    taps = np.hanning(257) / np.hanning(257).sum()

    def effects():
        return [F.LoButterworth(3000, order=4), StatefulFIR(taps)]
    whole = dev(torch.from_numpy(frames.T.copy()))
    fx = effects()
    fx[0].fs = 44100
    ref = fx[1](fx[0](whole)).cpu().numpy()
    for use_graph in (False, True):
        proc = StreamProcessor(effects(), chunk_size=8192, overlap=0, device=DEV, use_graph=use_graph)
        proc.process_file(src, tmp_path / "o" / "out.wav")
        rec = sf.written[-1]
        assert rec["format"] == "WAV" and rec["subtype"] == "FLOAT" and rec["fs"] == 44100 and rec["channels"] == 2
        assert rec["data"].shape == frames.shape
        assert np.abs(rec["data"].T - ref).max() <= 2e-6, use_graph
    chunks = list(StreamProcessor(effects(), chunk_size=8192, device=DEV).process_file_chunks(src))
    assert [c.shape[1] for c in chunks] == [8192] * 6 + [50_000 - 6 * 8192] and not chunks[0].is_cuda
    assert np.abs(torch.cat(chunks, dim=1).numpy() - ref).max() <= 2e-6


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("C,T,K,taps,gain,clamp", [
    (2, 512, 4, 301, 1.7, True), (1, 1, 1, 1, None, False), (3, 4096, 2, 256, 0.5, False), (2, 511, 3, 64, None, True),
    (5, 1000, 0, 129, 2.0, False), (2, 128, 6, 1, 0.25, True), (64, 4096, 4, 255, None, False), (2, 2048, 8, 512, -1.5, True), (2, 4096, 4, 1024, 1.0, False), (3, 1000, 2, 4096, None, False)])
def test_chunk_forward_equals_the_staged_ops(C, T, K, taps, gain, clamp, prec):
    """tfx_chunk_forward (one launch: cascade -> stateful direct FIR -> gain / clip) against the three staged ops on
    three consecutive chunks with carried state and history (same cascade code with shorter lane chunks, same float32
    FMA order as the plain direct kernel: equal to float64 round-off of the recursion), and against the oracle on the
    whole signal."""
    from scipy.signal import butter
    E = ext()
    g = np.random.default_rng(C * 1000 + T)
    sos = np.vstack([butter(2, (0.05 + 0.1 * i), output="sos") for i in range(K)]) if K else np.zeros((0, 6))
    kf = (g.standard_normal(taps) / max(1, taps) ** 0.5).astype(np.float32)
    x = rnd((C, 3 * T), C + T)
    xd = dev(x)
    sx = sy = hist = None
    ssx = ssy = shist = None
    outs = []
    for i in range(3):
        xv = xd[:, i * T:(i + 1) * T]                   # a column window of the longer buffer: taken by row pitch, no copy
        xc = xv.contiguous()
        y, sx, sy, hist = E.chunk_forward(xv if i else xc, torch.from_numpy(sos), sx, sy, torch.from_numpy(kf), hist, gain, clamp, precision=prec)
        u = xc
        if K:
            u, ssx, ssy = E.sos_forward(xc, None, torch.from_numpy(sos), ssx, ssy, precision=prec)
        if taps > 1:
            u, shist = E.fir_stream_forward(u, torch.from_numpy(kf), shist, True)
        else:
            u = u * float(kf[0])
        if gain is not None or clamp:
            u = E.gain_forward(u, 1.0 if gain is None else gain, clamp)
        # the staged cascade walks 64-sample lane chunks (LC 64 / 32), the chunk kernel 16-sample ones: the float64
        # recursions agree to round-off, so the float32 samples are equal except where 1e-16 crosses a rounding boundary
        ytol = 1.5e-7 if prec == "f64" else 2e-5         # float32 recursions with different lane chunks differ like any two float32 orders
        close(y, u.cpu().numpy(), ytol, f"chunk {i}: fused vs staged")
        if K:
            close(sx, ssx.cpu().numpy(), 1e-12 if prec == "f64" else 2e-5)
            close(sy, ssy.cpu().numpy(), 1e-12 if prec == "f64" else 2e-5)
        if taps > 1:
            close(hist, shist.cpu().numpy(), ytol)
        outs.append(y)
    if prec == "f64":
        ref = O.sos_forward(x, sos)[0].astype(np.float32) if K else x
        ref = O.fir_direct(ref, kf)
        if gain is not None:
            ref = ref * np.float32(gain)
        if clamp:
            ref = np.clip(ref, -1.0, 1.0)
        close(torch.cat(outs, dim=1), ref, 1e-5, "fused chunks vs oracle")


def test_stream_processor_takes_the_fused_chunk_path_on_device(monkeypatch):
    """LoButterworth-4 | ParametricEQ | StatefulFIR-301 | Gain(clamp) in 512-sample chunks: one chunk_iir_fir_kernel launch
    per chunk (counted by the library's profiler), equal to the one-shot oracle; TORCHFX_AMD_FUSE_CHUNK=0 gives the staged
    launches and the same samples to a float32 ulp."""
    import json
    from torchfx_amd import _lib
    from torchfx_amd import filter as F
    from torchfx_amd.effect import Gain
    from torchfx_amd.realtime import StatefulFIR, StreamProcessor
    lib = _lib.load()
    fs = 48000
    b = (np.hanning(301) / np.hanning(301).sum())
    x = rnd((2, 512 * 20), 5)

    def chain():
        return [F.LoButterworth(3000, order=4, fs=fs), F.ParametricEQ(frequency=800, q=2.0, gain=4.0, fs=fs),
                StatefulFIR(b, conv_mode="fft"), Gain(1.9, clamp=True)]
    sp = StreamProcessor(chain(), chunk_size=512, device=DEV)
    lib.tfx_prof_enable(1)
    lib.tfx_prof_collect()
    y = sp.process_tensor(torch.from_numpy(x), fs)
    torch.cuda.synchronize()
    prof = json.loads(lib.tfx_prof_collect().decode())
    lib.tfx_prof_enable(0)
    assert set(prof) == {"chunk_iir_fir_kernel"} and prof["chunk_iir_fir_kernel"]["calls"] == 20, prof
    mods = chain()
    for m in mods[:2]:
        m.compute_coefficients()
    sos = np.vstack([m._sos.numpy() for m in mods[:2]])
    ref = np.clip(O.fir_direct(O.sos_forward(x, sos)[0].astype(np.float32), b[::-1].astype(np.float32).copy()) * np.float32(1.9), -1, 1)
    close(y, ref, 1e-5, "fused chunks vs one-shot oracle")
    monkeypatch.setenv("TORCHFX_AMD_FUSE_CHUNK", "0")
    ys = StreamProcessor(chain(), chunk_size=512, device=DEV).process_tensor(torch.from_numpy(x), fs)
    close(y, ys.cpu().numpy(), 3e-7, "fused vs staged chunk loop")


def test_fused_chunk_run_rezeroes_state_when_the_channel_count_changes():
    """Advisor, round 3: a `_ChunkRun` that has run on C rows and then gets a chunk with another channel count (a processor
    reused on another signal) must re-zero the IIR state like `_sos_cascade_forward` and the reference do (iir.py:136-138,
    tests/test_fused.py:202-214 of the reference) instead of handing the kernel state tensors of the old shape."""
    from torchfx_amd import filter as F
    from torchfx_amd.effect import Gain
    from torchfx_amd.realtime import StatefulFIR, _ChunkRun
    fs = 48000
    b = np.hanning(65) / np.hanning(65).sum()
    iirs = [F.LoButterworth(3000, order=4, fs=fs), F.ParametricEQ(frequency=800, q=2.0, gain=4.0, fs=fs)]
    run = _ChunkRun(iirs, StatefulFIR(b, conv_mode="fft"), Gain(0.9))
    x2, x5 = rnd((2, 1024), 1), rnd((5, 1024), 2)
    assert run.fuses(dev(x2)) and run.fuses(dev(x5))
    run(dev(x2[:, :512]))
    run(dev(x2[:, 512:]))                                 # fast path: states carried as views of the combined tensors
    y5 = run(dev(x5[:, :512]))                            # other channel count: states and history start from zero again
    y5b = run(dev(x5[:, 512:]))
    for m in iirs:
        m.compute_coefficients()
    sos = np.vstack([m._sos.numpy() for m in iirs])
    ref = O.fir_direct(O.sos_forward(x5, sos)[0].astype(np.float32), b[::-1].astype(np.float32).copy()) * np.float32(0.9)
    close(torch.cat([y5, y5b], dim=1), ref, 1e-5, "5-channel signal after a 2-channel one == one shot from zero state")
    assert tuple(iirs[0]._state_x.shape)[1] == 5
