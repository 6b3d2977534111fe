"""ctypes host layer over libaccelgpu.so.

Method names mirror the reference functions they stand in for
(create_subharminfos, subharm_ffdot_plane, fund_to_ffdotplane,
inmem_add_ffdotpows, search_ffdotpows; reference include/accel.h:116-147).
There is no CPU fallback: if the library is missing, loading fails loudly.
"""
import collections
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libaccelgpu.so")
USELEN = 7470
MAXSTAGES = 6


class AccelGPUError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("accelgpu error %d: %s" % (code, msg))
        self.code = code


class Setup(C.Structure):
    _fields_ = [("N", C.c_int64), ("numbins", C.c_int64), ("lobin", C.c_int64),
                ("highestbin", C.c_int64),
                ("zmax", C.c_int32), ("numz", C.c_int32), ("numharm", C.c_int32),
                ("numharmstages", C.c_int32), ("fftlen", C.c_int32), ("halfwidth", C.c_int32),
                ("num_search_blocks", C.c_int32), ("num_prepop_blocks", C.c_int32),
                ("dt", C.c_double), ("T", C.c_double), ("rlo", C.c_double), ("rhi", C.c_double),
                ("zlo", C.c_double), ("zhi", C.c_double),
                ("sigma", C.c_float), ("powcut", C.c_float * MAXSTAGES),
                ("numindep", C.c_int64 * MAXSTAGES),
                ("rlen", C.c_int64), ("plane_bytes", C.c_int64)]


HIT_DTYPE = np.dtype([("power", "f4"), ("stage", "i4"), ("zind", "i4"), ("block", "i4"),
                      ("rind", "i8")])
CAND_DTYPE = np.dtype([("power", "f4"), ("sigma", "f4"), ("numharm", "i4"), ("added_order", "i4"),
                       ("r", "f8"), ("z", "f8")])

_lib = None


def build_library():
    """Compile libaccelgpu.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `make -C presto_amd/csrc` (or "
                          "__graft_entry__.build()).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, i64, d, f = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
    pS = C.POINTER(Setup)
    L.accelgpu_last_error.restype = C.c_char_p
    L.accelgpu_version.restype = C.c_char_p
    L.accelgpu_create.argtypes = [C.POINTER(vp), i]
    L.accelgpu_destroy.argtypes = [vp]; L.accelgpu_destroy.restype = None
    L.accelgpu_set_stream.argtypes = [vp, vp]
    L.accelgpu_synchronize.argtypes = [vp]
    L.accelgpu_bind_thread.argtypes = [vp]
    L.accelgpu_z_resp_halfwidth.argtypes = [d]
    L.accelgpu_calc_fftlen.argtypes = [i, i, i]
    L.accelgpu_coarse_bound_applies.argtypes = [i, i64, i64, i, i]
    L.accelgpu_block_range_need.argtypes = [pS, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.accelgpu_candidate_sigma.argtypes = [d, i, d]; L.accelgpu_candidate_sigma.restype = d
    L.accelgpu_power_for_sigma.argtypes = [d, i, d]; L.accelgpu_power_for_sigma.restype = d
    L.accelgpu_plan.argtypes = [i64, i64, d, i, i, d, d, d, i64, pS]
    L.accelgpu_build_kernels.argtypes = [vp, i]
    L.accelgpu_get_kernel.argtypes = [vp, i, vp]
    L.accelgpu_load_fft.argtypes = [vp, vp, i64]
    L.accelgpu_load_fft_device.argtypes = [vp, vp, i64]
    L.accelgpu_load_fft_async.argtypes = [vp, vp, i64]
    L.accelgpu_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.accelgpu_host_free.argtypes = [vp]
    L.accelgpu_set_tuning.argtypes = [vp, C.c_char_p, i]
    L.accelgpu_get_tuning.argtypes = [vp, C.c_char_p, C.POINTER(i)]
    L.accelgpu_load_dat.argtypes = [vp, vp, i64, i]
    L.accelgpu_fund_blocks_noalloc.argtypes = [vp, C.POINTER(d), i, vp, C.POINTER(C.c_int32)]
    L.accelgpu_deredden.argtypes = [vp]
    L.accelgpu_set_bin0.argtypes = [vp, f, f]
    L.accelgpu_rednoise.argtypes = [vp, i, i, d, d]
    L.accelgpu_get_fft.argtypes = [vp, vp, i64]
    L.accelgpu_zapbirds.argtypes = [vp, vp, vp, C.c_int32]
    L.accelgpu_read_birds.argtypes = [C.c_char_p, d, d, vp, vp, C.c_int32, C.POINTER(C.c_int32)]
    L.accelgpu_calc_props.argtypes = [vp, d, d, d, vp]
    L.accelgpu_nice_output.argtypes = [i, d, d, C.c_char_p, i]
    L.accelgpu_write_accel_files.argtypes = [pS, vp, i64, f, C.c_char_p, C.c_char_p, C.c_char_p]
    L.accelgpu_write_txtcand.argtypes = [pS, vp, vp, vp, i64, C.c_char_p]
    L.accelgpu_sort_and_eliminate.argtypes = [vp, i64, i, C.POINTER(i64)]
    L.accelgpu_optimize_cands.argtypes = [vp, vp, i64, i, vp]
    L.accelgpu_sift_records.argtypes = [vp, i64, f, vp]
    L.accelgpu_optimize_launch.argtypes = [vp, vp, i64, i]
    L.accelgpu_optimize_collect.argtypes = [vp, vp, i64]
    L.accelgpu_set_search.argtypes = [vp, pS, i, f]
    L.accelgpu_set_block_range.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.accelgpu_fill_plane.argtypes = [vp]
    L.accelgpu_search_plane.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.accelgpu_run.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.accelgpu_run_launch.argtypes = [vp]
    L.accelgpu_run_collect.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.accelgpu_replay_candidates.argtypes = [pS, vp, i64, vp, i64, C.POINTER(i64), vp, vp]
    L.accelgpu_replay_options.argtypes = [i, i]
    L.accelgpu_fund_block.argtypes = [vp, d, vp, C.POINTER(C.c_int32)]
    L.accelgpu_summed_block.argtypes = [vp, i, i, vp]
    p32 = C.POINTER(C.c_int32)
    L.accelgpu_set_search_standard.argtypes = [vp, pS, i, f]
    L.accelgpu_subharm_ffdot_plane.argtypes = [vp, i, i, d, d, vp, i64, p32, p32, p32, p32, vp]
    L.accelgpu_add_ffdotpows.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, i, i]
    L.accelgpu_run_standard.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.accelgpu_get_plane.argtypes = [vp, i, i64, i64, vp]
    L.accelgpu_put_plane.argtypes = [vp, i, i64, i64, vp]
    L.accelgpu_set_profiling.argtypes = [vp, i]
    L.accelgpu_get_timings.argtypes = [vp, C.POINTER(i64), C.POINTER(d), i]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise AccelGPUError(rc, load_library().accelgpu_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def plan(numbins, N, dt, zmax, numharm, sigma=2.0, flo=1.0, fhi=10000.0, lobin=0):
    """Search set-up as create_accelobs() derives it (no GPU needed)."""
    s = Setup()
    _check(load_library().accelgpu_plan(int(numbins), int(N), float(dt), int(zmax), int(numharm),
                                        float(sigma), float(flo), float(fhi), int(lobin),
                                        C.byref(s)))
    return s


def candidate_sigma(power, numsum, numtrials):
    return load_library().accelgpu_candidate_sigma(float(power), int(numsum), float(numtrials))


def power_for_sigma(sigma, numsum, numtrials):
    return load_library().accelgpu_power_for_sigma(float(sigma), int(numsum), float(numtrials))


MAXHARM = 32
RDERIVS_DTYPE = np.dtype([("pow", "f8"), ("phs", "f8"), ("dpow", "f8"), ("dphs", "f8"),
                          ("d2pow", "f8"), ("d2phs", "f8"), ("locpow", "f8")])
OPTCAND_DTYPE = np.dtype([("power", "f4"), ("sigma", "f4"), ("numharm", "i4"), ("_pad", "i4"),
                          ("r", "f8"), ("z", "f8"), ("pows", "f8", MAXHARM), ("hirs", "f8", MAXHARM),
                          ("hizs", "f8", MAXHARM), ("derivs", RDERIVS_DTYPE, MAXHARM)])


FOURIERPROPS_DTYPE = np.dtype([("r", "f8"), ("rerr", "f4"), ("z", "f8"), ("zerr", "f4"), ("w", "f8"),
                               ("werr", "f4"), ("pow", "f4"), ("powerr", "f4"), ("sig", "f4"),
                               ("rawpow", "f4"), ("phs", "f4"), ("phserr", "f4"), ("cen", "f4"),
                               ("cenerr", "f4"), ("pur", "f4"), ("purerr", "f4"), ("locpow", "f4")],
                              align=True)


SIFTREC_DTYPE = np.dtype([("sigma", "f8"), ("ipow", "f8"), ("cpow", "f8"), ("r", "f8"), ("z", "f8"),
                          ("numharm", "i4"), ("_pad", "i4"), ("harm_pow", "f8", MAXHARM),
                          ("harm_phs", "f8", MAXHARM)])


def sift_records(optcands, nph=0.0):
    """Per optimised candidate, the numbers the sifting step would read back from the
    ACCEL file (printed precision): SIFTREC_DTYPE records."""
    c = np.ascontiguousarray(optcands, dtype=OPTCAND_DTYPE)
    out = np.zeros(c.size, dtype=SIFTREC_DTYPE)
    _check(load_library().accelgpu_sift_records(_ptr(c), c.size, float(nph), _ptr(out)))
    return out


def nice_output(error_digits, val, err):
    """value(error) string with 1 or 2 error digits (nice_output_1 / nice_output_2)."""
    buf = C.create_string_buffer(80)
    _check(load_library().accelgpu_nice_output(int(error_digits), float(val), float(err), buf, 80))
    return buf.value.decode()


def calc_props(derivs, r, z, w=0.0):
    d = np.ascontiguousarray(derivs, dtype=RDERIVS_DTYPE).reshape(1)
    out = np.zeros(1, dtype=FOURIERPROPS_DTYPE)
    _check(load_library().accelgpu_calc_props(_ptr(d), float(r), float(z), float(w), _ptr(out)))
    return out[0]


def write_accel_files(setup, optcands, accel_path, cand_path, inf_text=None, nph=0.0):
    """The ACCEL text file (+ .inf text appended) and the .cand file of fourierprops."""
    c = np.ascontiguousarray(optcands, dtype=OPTCAND_DTYPE)
    _check(load_library().accelgpu_write_accel_files(
        C.byref(setup), _ptr(c), c.size, float(nph), accel_path.encode(), cand_path.encode(),
        None if inf_text is None else inf_text.encode()))


def read_birds(path, T, baryv=0.0, max_birds=4096):
    """Birdie file -> (lobins, hibins) in Fourier bins, sorted (get_birdies)."""
    lo = np.zeros(max_birds); hi = np.zeros(max_birds)
    n = C.c_int32(0)
    _check(load_library().accelgpu_read_birds(path.encode(), float(T), float(baryv), _ptr(lo), _ptr(hi),
                                              max_birds, C.byref(n)))
    return lo[:n.value].copy(), hi[:n.value].copy()


def sort_and_eliminate(cands, eliminate=True):
    """sort_accelcands + eliminate_harmonics on the host: the kept candidates."""
    c = np.ascontiguousarray(cands, dtype=CAND_DTYPE).copy()
    n = C.c_int64(0)
    _check(load_library().accelgpu_sort_and_eliminate(_ptr(c), c.size, int(bool(eliminate)), C.byref(n)))
    return c[:n.value].copy()


def replay_options(linear_walk=False, threads=-1):
    """How replay_candidates() works (never what it returns): the reference's plain list
    walk instead of the binary search; a fixed number of sigma threads."""
    _check(load_library().accelgpu_replay_options(int(bool(linear_walk)), int(threads)))


def block_range_need(setup, first_block, num_blocks):
    """(plane blocks, (block, z row)s) a rank that searches the search blocks [first, first + n) of `setup`
    has to generate.  Host arithmetic only."""
    L = load_library()
    nb, rows = C.c_int32(0), C.c_int64(0)
    _check(L.accelgpu_block_range_need(C.byref(setup), int(first_block), int(num_blocks), C.byref(nb), C.byref(rows)))
    return nb.value, rows.value


def write_txtcand(setup, hits, sigma, added, path):
    """The search's work file from what replay_candidates() returned (one C call; it releases the interpreter lock)."""
    hits = np.ascontiguousarray(hits)
    sigma = np.ascontiguousarray(sigma, dtype=np.float32)
    added = np.ascontiguousarray(added, dtype=np.int32)
    _check(load_library().accelgpu_write_txtcand(C.byref(setup), hits.ctypes.data, sigma.ctypes.data, added.ctypes.data,
                                                 len(hits), path.encode()))


def replay_candidates(setup, hits):
    """Host replay: returns (cands, sigma per hit, added flag per hit)."""
    L = load_library()
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    n = hits.size
    cap = max(n, 1)
    cands = np.zeros(cap, dtype=CAND_DTYPE)
    sig = np.zeros(cap, dtype=np.float32)
    added = np.zeros(cap, dtype=np.int32)
    nc = C.c_int64(0)
    _check(L.accelgpu_replay_candidates(C.byref(setup), _ptr(hits), n, _ptr(cands), cap,
                                        C.byref(nc), _ptr(sig), _ptr(added)))
    return cands[:nc.value].copy(), sig[:n], added[:n]


class PinnedSpectrum:
    """complex64[numbins] in page-locked host memory (accelgpu_host_alloc): what
    AccelSearchGPU.load_fft_async() uploads from without the host waiting."""

    def __init__(self, numbins):
        self._p = C.c_void_p()
        _check(load_library().accelgpu_host_alloc(C.byref(self._p), int(numbins) * 8))
        buf = (C.c_float * (2 * int(numbins))).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=np.complex64)

    def free(self):
        if self._p:
            self.array = None
            load_library().accelgpu_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class AccelSearchGPU:
    """One search context on one GPU."""

    def __init__(self, device=0):
        self.L = load_library()
        self._h = C.c_void_p()
        _check(self.L.accelgpu_create(C.byref(self._h), int(device)))
        self.setup = None
        # objects that own the memory queued work reads: the spectrum loaded for the next
        # launch, one per launch in flight, and the one of the search collected last
        # (which a launched optimisation may still read)
        self._keep_next = None
        self._keep_flight = collections.deque()
        self._keep_collected = None

    def close(self):
        if self._h:
            self.L.accelgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- set-up -----------------------------------------------------------
    def set_stream(self, hip_stream):
        _check(self.L.accelgpu_set_stream(self._h, C.c_void_p(hip_stream)))

    def create_subharminfos(self, zmax):
        """Build (or reuse) the kernel bank for the fundamental."""
        _check(self.L.accelgpu_build_kernels(self._h, int(zmax)))

    def kernel(self, zind, fftlen):
        out = np.empty(fftlen, dtype=np.complex64)
        _check(self.L.accelgpu_get_kernel(self._h, int(zind), _ptr(out)))
        return out

    def load_fft(self, spectrum):
        a = np.ascontiguousarray(spectrum, dtype=np.complex64)
        _check(self.L.accelgpu_load_fft(self._h, _ptr(a), a.size))

    def load_dat(self, series, deredden=True):
        """Time series in: realfft (+ deredden) on the device; returns numbins."""
        a = np.ascontiguousarray(series, dtype=np.float32)
        _check(self.L.accelgpu_load_dat(self._h, _ptr(a), a.size, int(bool(deredden))))
        self._numbins = a.size // 2
        return self._numbins

    def set_bin0(self, re, im):
        """Overwrite bin 0 of the current spectrum (the reference sets (1, 1) for
        time-series input without -photon, src/accel_utils.c:1553-1556)."""
        _check(self.L.accelgpu_set_bin0(self._h, float(re), float(im)))

    def deredden(self):
        _check(self.L.accelgpu_deredden(self._h))

    def rednoise(self, T, startwidth=6, endwidth=100, endfreq=6.0):
        _check(self.L.accelgpu_rednoise(self._h, int(startwidth), int(endwidth), float(endfreq), float(T)))

    def zapbirds(self, lobins, hibins):
        lo = np.ascontiguousarray(lobins, dtype=np.float64)
        hi = np.ascontiguousarray(hibins, dtype=np.float64)
        _check(self.L.accelgpu_zapbirds(self._h, _ptr(lo), _ptr(hi), lo.size))

    def get_fft(self, numbins):
        out = np.empty(int(numbins), dtype=np.complex64)
        _check(self.L.accelgpu_get_fft(self._h, _ptr(out), out.size))
        return out

    def load_fft_device(self, dev_ptr, numbins, keepalive=None):
        """Use a spectrum already in device memory (e.g. a torch tensor's data_ptr());
        `keepalive` (the tensor) is held until the search launched on it has been
        collected and the next one after it too (its optimisation may still read it)."""
        _check(self.L.accelgpu_load_fft_device(self._h, C.c_void_p(dev_ptr), int(numbins)))
        self._keep_next = keepalive

    def load_fft_async(self, pinned):
        """Upload from a PinnedSpectrum without waiting; the launch waits on the device."""
        _check(self.L.accelgpu_load_fft_async(self._h, C.c_void_p(pinned._p.value), pinned.array.size))
        self._keep_next = pinned

    def bind_thread(self):
        """Makes the context's GPU the current device of the calling thread (for a helper thread that
        allocates page-locked memory)."""
        _check(self.L.accelgpu_bind_thread(self._h))

    def set_tuning(self, name, value):
        _check(self.L.accelgpu_set_tuning(self._h, name.encode(), int(value)))

    def get_tuning(self, name):
        v = C.c_int(0)
        _check(self.L.accelgpu_get_tuning(self._h, name.encode(), C.byref(v)))
        return v.value

    def configure(self, numbins, N, dt, zmax, numharm, sigma=2.0, flo=1.0, fhi=10000.0, lobin=0,
                  photon_nph=0.0, norm_type=0, inmem=True):
        """plan + kernel bank + set_search in one call.  inmem=False selects the
        reference's standard mode (no plane; see run_standard)."""
        setup = plan(numbins, N, dt, zmax, numharm, sigma, flo, fhi, lobin)
        self.create_subharminfos(setup.zmax)
        fn = self.L.accelgpu_set_search if inmem else self.L.accelgpu_set_search_standard
        _check(fn(self._h, C.byref(setup), int(norm_type), float(photon_nph)))
        self.setup = setup                           # only once the library has accepted it
        return self.setup

    def set_search(self, setup, norm_type=0, photon_nph=0.0, inmem=True):
        """accelgpu_set_search() with a set-up the caller filled or changed (thresholds of its
        own, say); the kernel bank for setup.zmax must exist (create_subharminfos)."""
        fn = self.L.accelgpu_set_search if inmem else self.L.accelgpu_set_search_standard
        _check(fn(self._h, C.byref(setup), int(norm_type), float(photon_nph)))
        self.setup = setup
        return self.setup

    def set_block_range(self, first_block, num_blocks):
        """Search only blocks [first, first+num) of the grid (one long series split
        over GPUs).  Returns the number of plane blocks a fill will generate."""
        n = C.c_int32(0)
        _check(self.L.accelgpu_set_block_range(self._h, int(first_block), int(num_blocks), C.byref(n)))
        return n.value

    # -- the hot path -----------------------------------------------------
    def fill_plane(self):
        _check(self.L.accelgpu_fill_plane(self._h))

    def _hits_call(self, fn, max_hits):
        while True:
            hits = np.empty(max_hits, dtype=HIT_DTYPE)
            n = C.c_int64(0)
            rc = fn(self._h, _ptr(hits), max_hits, C.byref(n))
            if rc == -4 and n.value > max_hits:      # ACCELGPU_ERR_OVERFLOW: grow and repeat
                max_hits = int(n.value * 1.25) + 1024
                continue
            _check(rc)
            return hits[:n.value]

    def search_plane(self, max_hits=1 << 20):
        return self._hits_call(self.L.accelgpu_search_plane, max_hits)

    def run(self, max_hits=1 << 20):
        self.run_launch()
        return self.run_collect(max_hits)

    def run_one_call(self, max_hits=1 << 20):
        """accelgpu_run() itself (fill + search in one C call, repeated from scratch with a
        larger buffer when the hits do not fit)."""
        return self._hits_call(self.L.accelgpu_run, max_hits)

    def run_launch(self):
        _check(self.L.accelgpu_run_launch(self._h))
        self._keep_flight.append(self._keep_next)

    def run_collect(self, max_hits=1 << 20):
        hits = self._hits_call(self.L.accelgpu_run_collect, max_hits)
        if self._keep_flight:
            self._keep_collected = self._keep_flight.popleft()
        return hits

    def search(self, max_hits=1 << 20):
        """Whole search of the loaded spectrum: (hits, cands, sigma, added)."""
        hits = self.run(max_hits)
        cands, sig, added = replay_candidates(self.setup, hits)
        return hits, cands, sig, added

    # -- after the search -----------------------------------------------------
    def optimize(self, cands, polish=True):
        """optimize_accelcand for every candidate on the GPU: OPTCAND_DTYPE records."""
        c = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        out = np.zeros(c.size, dtype=OPTCAND_DTYPE)
        _check(self.L.accelgpu_optimize_cands(self._h, _ptr(c), c.size, int(bool(polish)), _ptr(out)))
        return out

    def optimize_launch(self, cands, polish=True):
        """First half of optimize(): queue the device work and return."""
        c = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        _check(self.L.accelgpu_optimize_launch(self._h, _ptr(c), c.size, int(bool(polish))))
        return c.size

    def optimize_collect(self, n):
        """Second half: wait and return the n OPTCAND_DTYPE records."""
        out = np.zeros(int(n), dtype=OPTCAND_DTYPE)
        _check(self.L.accelgpu_optimize_collect(self._h, _ptr(out), out.size))
        return out

    # -- standard (not in-memory) mode ---------------------------------------
    def run_standard(self, max_hits=1 << 20):
        """Whole main loop without a plane: sorted hits."""
        return self._hits_call(self.L.accelgpu_run_standard, max_hits)

    def search_standard(self, max_hits=1 << 20):
        hits = self.run_standard(max_hits)
        cands, sig, added = replay_candidates(self.setup, hits)
        return hits, cands, sig, added

    def subharm_block(self, numharm, harmnum, fullrlo, fullrhi=None):
        """subharm_ffdot_plane() for any harmonic:
        (powers[numzs, numrs], rlo, zlo, rinds[USELEN] uint16)."""
        if fullrhi is None:
            fullrhi = fullrlo + USELEN * 0.5 - 0.5
        cap = self.setup.numz * (USELEN + 2)
        buf = np.empty(cap, dtype=np.float32)
        rinds = np.zeros(USELEN, dtype=np.uint16)
        rlo, zlo, numrs, numzs = (C.c_int32(0) for _ in range(4))
        _check(self.L.accelgpu_subharm_ffdot_plane(
            self._h, int(numharm), int(harmnum), float(fullrlo), float(fullrhi), _ptr(buf), cap,
            C.byref(rlo), C.byref(zlo), C.byref(numrs), C.byref(numzs), _ptr(rinds)))
        pw = buf[:numzs.value * numrs.value].reshape(numzs.value, numrs.value).copy()
        return pw, rlo.value, zlo.value, rinds

    def add_ffdotpows(self, fund, sub, sub_zlo, rinds, numharm, harmnum):
        """fund[z][j] += sub[zind(z)][rinds[j]] in place (fund: float32 [numz, USELEN])."""
        assert fund.dtype == np.float32 and fund.flags.c_contiguous
        sub = np.ascontiguousarray(sub, dtype=np.float32)
        rinds = np.ascontiguousarray(rinds, dtype=np.uint16)
        _check(self.L.accelgpu_add_ffdotpows(self._h, _ptr(fund), _ptr(sub), int(sub_zlo),
                                             sub.shape[1], sub.shape[0], _ptr(rinds),
                                             int(numharm), int(harmnum)))

    # -- parity getters -----------------------------------------------------
    def subharm_ffdot_plane(self, startr):
        """Fundamental block at startr: (powers[numz, USELEN], rlo)."""
        out = np.empty((self.setup.numz, USELEN), dtype=np.float32)
        rlo = C.c_int32(0)
        _check(self.L.accelgpu_fund_block(self._h, float(startr), _ptr(out), C.byref(rlo)))
        return out, rlo.value

    def fund_blocks_noalloc(self, startrs, out_dev_ptr):
        """Fundamental blocks at startrs into a caller's device buffer
        ([len(startrs), numz, USELEN] float32) without allocating or waiting; returns
        each block's rlo.  Synchronise the context before reading the buffer."""
        sr = (C.c_double * len(startrs))(*[float(x) for x in startrs])
        rlo = (C.c_int32 * len(startrs))()
        _check(self.L.accelgpu_fund_blocks_noalloc(self._h, sr, len(startrs), C.c_void_p(out_dev_ptr), rlo))
        return list(rlo)

    def summed_block(self, block, stage):
        out = np.empty((self.setup.numz, USELEN), dtype=np.float32)
        _check(self.L.accelgpu_summed_block(self._h, int(block), int(stage), _ptr(out)))
        return out

    def get_plane(self, zind, col, count):
        out = np.empty(count, dtype=np.float32)
        _check(self.L.accelgpu_get_plane(self._h, int(zind), int(col), int(count), _ptr(out)))
        return out

    def put_plane(self, zind, col, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        _check(self.L.accelgpu_put_plane(self._h, int(zind), int(col), v.size, _ptr(v)))

    def plane(self):
        rows = [self.get_plane(z, 0, self.setup.rlen) for z in range(self.setup.numz)]
        return np.stack(rows)

    # -- measurement --------------------------------------------------------
    def set_profiling(self, on):
        _check(self.L.accelgpu_set_profiling(self._h, int(bool(on))))

    def timings(self, reset=True):
        n = (C.c_int64 * 3)()
        ms = (C.c_double * 3)()
        _check(self.L.accelgpu_get_timings(self._h, n, ms, int(bool(reset))))
        names = ("block_prep", "correlate", "harmsum")
        return {k: dict(launches=int(n[i]), total_ms=float(ms[i])) for i, k in enumerate(names)}

    def synchronize(self):
        _check(self.L.accelgpu_synchronize(self._h))
