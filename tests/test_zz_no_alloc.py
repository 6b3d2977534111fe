"""The entry point that allocates nothing, accelgpu_fund_blocks_noalloc(), against the
entry point it is a variant of, accelgpu_fund_block(): same spectrum, same blocks, bit
for bit (the same kernels do the same arithmetic; only where the block spectra are
staged differs), on two calls in a row, the second one smaller than the first.

Collected last on purpose (file name): a failure here must not hide the other tests
from a run that stops at the first failure."""
import numpy as np
import pytest

from presto_amd import synth

pytestmark = pytest.mark.gpu

USELEN = 7470


@pytest.mark.parametrize("name,N,zmax,numharm", [("cfg2", 1 << 19, 200, 16), ("cfg5", 1 << 19, 1200, 32)])
def test_no_alloc_blocks_equal_fund_block(name, N, zmax, numharm):
    import torch
    import presto_amd as pa
    fft, c = synth.config_fft(name, N=N)
    g = pa.AccelSearchGPU(0)
    try:
        setup = g.configure(fft.size, N, c["dt"], zmax, numharm, fhi=1e9)
        g.load_fft(fft)
        numz = setup.numz
        rlo = setup.rlo
        last = rlo + 3735.0 * (setup.num_search_blocks - 1)
        first_call = [8.0, rlo, rlo + 3735.0, rlo + 2 * 3735.0, last]
        second_call = [last, rlo + 3735.0]
        want = {s: g.subharm_ffdot_plane(s) for s in set(first_call + second_call)}
        # the caller's buffer: torch is only the allocator here; poisoned so that a cell the
        # call does not write shows up
        out = torch.full((len(first_call), numz, USELEN), float("nan"), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        for starts in (first_call, second_call):
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            rlos = g.fund_blocks_noalloc(starts, out.data_ptr())
            g.synchronize()
            got = out.cpu().numpy()
            for b, s in enumerate(starts):
                assert rlos[b] == want[s][1]
                assert np.array_equal(got[b], want[s][0]), (s, b)
            assert np.isnan(got[len(starts):]).all()      # nothing written past the blocks asked for
        # more blocks than the fixed working memory holds: refused, with the reason
        with pytest.raises(pa.AccelGPUError, match="holds 16"):
            g.fund_blocks_noalloc([float(rlo)] * 17, out.data_ptr())
        # a host pointer as output: refused
        host = np.empty((1, numz, USELEN), dtype=np.float32)
        with pytest.raises(pa.AccelGPUError, match="device memory"):
            g.fund_blocks_noalloc([float(rlo)], host.ctypes.data)
    finally:
        g.close()
