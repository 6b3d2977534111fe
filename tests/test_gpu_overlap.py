"""Overlapped use of one context: the next trial is uploaded, prepared and launched while
a search is in flight, and the candidates of the search collected last are optimised
against their own spectrum beside the queued searches.  Everything must equal the plain
one-after-the-other sequence bit for bit (same kernels, same inputs)."""
import numpy as np
import pytest

import presto_amd as pa
from presto_amd import synth
from presto_amd.api import replay_candidates, sort_and_eliminate

pytestmark = pytest.mark.gpu
N, DT, ZMAX, NH = 1 << 20, 1e-4, 20, 8


@pytest.fixture(scope="module")
def gpu():
    # torch (the allocator of the caller-owned device buffers below) first: its wheel brings
    # its own copy of the HIP runtime, and initialising that one after this library's has
    # been working for a while was seen to fail ("No HIP GPUs are available")
    import torch
    torch.cuda.init()
    g = pa.AccelSearchGPU(0)
    yield g
    g.close()


def _spectra():
    out = []
    for k, (f, z) in enumerate([(123.456, 3.0), (77.7, -11.0), (301.3, 0.0), (19.19, 8.0)]):
        out.append(synth.make_fft(N, DT, [dict(f=f, z=z, amp=0.25, duty=0.08)], seed=70 + k))
    return out


def _plain(gpu, setup, spec, birds=None):
    gpu.load_fft(spec)
    if birds:
        gpu.zapbirds(*birds)
    hits = gpu.run()
    cands, _, _ = replay_candidates(setup, hits)
    kept = sort_and_eliminate(cands)
    return hits, kept, gpu.optimize(kept)


def test_overlapped_sequence_equals_plain_sequence(gpu):
    specs = _spectra()
    setup = gpu.configure(N // 2, N, DT, ZMAX, NH, fhi=4000.0)
    birds = (np.array([5000.0, 12000.0]), np.array([5010.0, 12040.0]))
    want = [_plain(gpu, setup, s, birds if k == 1 else None) for k, s in enumerate(specs)]

    got = []
    gpu.load_fft(specs[0])
    gpu.run_launch()
    pending = None
    for k in range(len(specs)):
        if k + 1 < len(specs):                       # next trial in while trial k is searched
            gpu.load_fft(specs[k + 1])
            if k + 1 == 1:
                gpu.zapbirds(*birds)
            gpu.run_launch()
        hits = gpu.run_collect()
        cands, _, _ = replay_candidates(setup, hits)
        kept = sort_and_eliminate(cands)
        if pending is not None:
            got[-1].append(gpu.optimize_collect(pending))
        pending = gpu.optimize_launch(kept)          # against trial k's spectrum, trial k+1 in flight
        got.append([hits, kept])
    got[-1].append(gpu.optimize_collect(pending))

    for (h0, k0, o0), (h1, k1, o1) in zip(want, got):
        assert len(h0) > 0 and np.array_equal(h0, h1)
        assert np.array_equal(k0, k1)
        assert len(o0) > 0 and o0.tobytes() == o1.tobytes()


def test_blocking_optimize_with_a_search_in_flight(gpu):
    specs = _spectra()
    setup = gpu.configure(N // 2, N, DT, ZMAX, NH, fhi=4000.0)
    h0, k0, o0 = _plain(gpu, setup, specs[0])
    gpu.load_fft(specs[0])
    gpu.run_launch()
    gpu.load_fft(specs[1])
    gpu.run_launch()
    hits = gpu.run_collect()
    assert np.array_equal(hits, h0)
    assert gpu.optimize(k0).tobytes() == o0.tobytes()       # specs[1] is loaded and in flight
    gpu.run_collect()


def test_misuse_is_refused(gpu):
    specs = _spectra()
    setup = gpu.configure(N // 2, N, DT, ZMAX, NH, fhi=4000.0)
    h0, k0, o0 = _plain(gpu, setup, specs[0])
    gpu.load_fft(specs[0])
    gpu.run_launch()
    # a search is in flight and this context has collected none since: which spectrum?
    gpu.load_fft(specs[1])
    gpu.run_launch()
    gpu.run_collect()                                       # specs[0] collected, specs[1] in flight
    n = gpu.optimize_launch(k0)
    with pytest.raises(pa.AccelGPUError):
        gpu.optimize_launch(k0)                             # one optimisation at a time
    with pytest.raises(pa.AccelGPUError):
        gpu.optimize_collect(n + 1)                         # as many as were launched
    assert gpu.optimize_collect(n).tobytes() == o0.tobytes()
    with pytest.raises(pa.AccelGPUError):
        gpu.optimize_collect(n)                             # nothing launched
    with pytest.raises(pa.AccelGPUError):
        gpu.zapbirds(np.array([10.0]), np.array([12.0]))    # would change what a launched search reads
    gpu.load_fft(specs[2])
    gpu.run_launch()                                        # specs[1], specs[2] in flight
    with pytest.raises(pa.AccelGPUError):
        gpu.run_launch()                                    # two at most
    with pytest.raises(pa.AccelGPUError, match="collect the 2 launched"):
        gpu.run_one_call()                                  # its collect would return specs[1]'s hits
    with pytest.raises(pa.AccelGPUError, match="collect the 2 launched"):
        gpu.configure(N // 2, N, DT, ZMAX, NH, fhi=2000.0)  # queued trials belong to their geometry
    with pytest.raises(pa.AccelGPUError, match="collect the 2 launched"):
        gpu.create_subharminfos(ZMAX + 10)
    gpu.load_fft(specs[3])                                  # takes the buffer of the collected specs[0]
    with pytest.raises(pa.AccelGPUError):
        gpu.optimize(k0)                                    # its spectrum is gone
    gpu.run_collect()
    gpu.run_collect()
    # drained: everything works again
    h3, k3, o3 = _plain(gpu, setup, specs[3])
    assert len(h3) > 0


def test_reload_keeps_the_spectrum_of_a_queued_optimisation(gpu):
    specs = _spectra()
    setup = gpu.configure(N // 2, N, DT, ZMAX, NH, fhi=4000.0)
    h0, k0, o0 = _plain(gpu, setup, specs[0])
    gpu.load_fft(specs[0])
    gpu.run_launch()
    gpu.run_launch()                                        # the same spectrum twice
    gpu.run_collect()
    n = gpu.optimize_launch(k0)                             # beside the second search
    gpu.run_collect()
    gpu.load_fft(specs[1])                                  # nothing in flight: must still not land on specs[0]
    gpu.load_fft(specs[2])
    assert gpu.optimize_collect(n).tobytes() == o0.tobytes()
    h2, k2, o2 = _plain(gpu, setup, specs[2])
    hits = gpu.run()
    assert np.array_equal(hits, h2)


def test_device_hit_buffer_overflow_with_a_later_trial_queued(gpu):
    """A trial with more hits than the device hit buffer holds, collected after the next
    trial has been launched (which has overwritten the plane): the collect re-runs that
    trial on its own spectrum into a larger buffer, and both trials come back exactly as
    a plain one-after-the-other sequence returns them; so does a too-small caller buffer."""
    specs = _spectra()[:3]
    setup = gpu.configure(N // 2, N, DT, ZMAX, NH, fhi=4000.0)
    want = []
    for s in specs:
        gpu.load_fft(s)
        want.append(gpu.run())
    assert min(len(w) for w in want) > 64
    try:
        gpu.set_tuning("hits_cap", 64)
        assert gpu.get_tuning("hits_cap") == 64
        gpu.load_fft(specs[0])
        gpu.run_launch()
        gpu.load_fft(specs[1])
        gpu.run_launch()
        got0 = gpu.run_collect(max_hits=8)       # device overflow and caller overflow at once
        gpu.load_fft(specs[2])
        gpu.run_launch()
        got1 = gpu.run_collect()
        got2 = gpu.run_collect()
        for g, w in zip((got0, got1, got2), want):
            assert np.array_equal(g, w)
        with pytest.raises(pa.AccelGPUError, match="nothing launched"):
            gpu.run_collect()
        # the whole-search call with a caller buffer that is too small leaves nothing queued
        gpu.load_fft(specs[0])
        assert np.array_equal(gpu.run_one_call(max_hits=4), want[0])
        with pytest.raises(pa.AccelGPUError, match="nothing launched"):
            gpu.run_collect()
    finally:
        gpu.set_tuning("hits_cap", 1 << 20)


def test_graph_launch_equals_plain_launch(gpu):
    """Small searches (plane <= 64 MB) are launched as one captured graph per (hit slot,
    spectrum buffer): same hits as the plain stream path, pipelined two deep, from library
    buffers, caller's device buffers and page-locked uploads, through a hit-buffer overflow
    (which drops the captured graphs) and across a change of geometry."""
    import torch
    from presto_amd.api import PinnedSpectrum
    g = pa.AccelSearchGPU(0)
    try:
        specs = _spectra()
        for zmax, nh in ((6, 8), (0, 4)):
            setup = g.configure(N // 2, N, DT, zmax, nh, fhi=4000.0)
            assert setup.plane_bytes <= 64 << 20
            g.set_tuning("use_graph", 0)
            want = []
            for s in specs:
                g.load_fft(s)
                want.append(g.run())
            assert min(len(w) for w in want) > 100
            g.set_tuning("use_graph", 1)

            def pipelined(load):
                got = []
                load(0)
                g.run_launch()
                load(1)
                g.run_launch()
                for k in range(2 * len(specs)):
                    got.append(g.run_collect())
                    if k + 2 < 2 * len(specs):
                        load(k + 2)
                        g.run_launch()
                return got

            dev = [torch.from_numpy(s.view(np.float32).copy()).to("cuda:0") for s in specs]
            pinned = []
            for s in specs:
                p = PinnedSpectrum(s.size)
                p.array[:] = s
                pinned.append(p)
            loaders = {
                "library buffers": lambda k: g.load_fft(specs[k % len(specs)]),
                "caller's device buffers": lambda k: g.load_fft_device(dev[k % len(specs)].data_ptr(), N // 2,
                                                                       keepalive=dev[k % len(specs)]),
                "page-locked uploads": lambda k: g.load_fft_async(pinned[k % len(specs)]),
            }
            for name, load in loaders.items():
                for k, hits in enumerate(pipelined(load)):
                    assert np.array_equal(hits, want[k % len(specs)]), (zmax, name, k)
            # overflow of the device hit buffer: re-run inside the collect, graphs dropped and rebuilt
            g.set_tuning("hits_cap", 64)
            for k, hits in enumerate(pipelined(loaders["library buffers"])):
                assert np.array_equal(hits, want[k % len(specs)]), (zmax, "small hit buffer", k)
            g.set_tuning("hits_cap", 1 << 20)
            for p in pinned:
                p.free()
    finally:
        g.close()
