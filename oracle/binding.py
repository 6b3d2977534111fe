"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: import this only from tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py.  The product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
USELEN = 7470
MAXSTAGES = 6


class Hit(C.Structure):
    _fields_ = [("power", C.c_float), ("sigma", C.c_float), ("numharm", C.c_int),
                ("block", C.c_int), ("zind", C.c_int), ("added", C.c_int),
                ("rind", C.c_longlong), ("r", C.c_double), ("z", C.c_double)]


class Cand(C.Structure):
    _fields_ = [("power", C.c_float), ("sigma", C.c_float), ("numharm", C.c_int),
                ("r", C.c_double), ("z", C.c_double)]


class Obs(C.Structure):
    _fields_ = [("N", C.c_longlong), ("numbins", C.c_longlong), ("lobin", C.c_longlong),
                ("highestbin", C.c_longlong),
                ("fftlen", C.c_int), ("numharmstages", C.c_int), ("numz", C.c_int),
                ("zmax", C.c_int), ("norm_type", C.c_int), ("inmem", C.c_int),
                ("dt", C.c_double), ("T", C.c_double), ("rlo", C.c_double), ("rhi", C.c_double),
                ("zlo", C.c_double), ("zhi", C.c_double),
                ("nph", C.c_float), ("sigma", C.c_float),
                ("powcut", C.c_float * MAXSTAGES), ("numindep", C.c_longlong * MAXSTAGES),
                ("fft", C.c_void_p), ("ffdotplane", C.POINTER(C.c_float)), ("rlen", C.c_longlong),
                ("halfwidth", C.c_int), ("kern", C.c_void_p), ("nthreads", C.c_int)]


MAXHARM = 32
RDERIVS_DTYPE = np.dtype([("pow", "f8"), ("phs", "f8"), ("dpow", "f8"), ("dphs", "f8"),
                          ("d2pow", "f8"), ("d2phs", "f8"), ("locpow", "f8")])
OPTCAND_DTYPE = np.dtype([("power", "f4"), ("sigma", "f4"), ("numharm", "i4"), ("_pad", "i4"),
                          ("r", "f8"), ("z", "f8"), ("pows", "f8", MAXHARM), ("hirs", "f8", MAXHARM),
                          ("hizs", "f8", MAXHARM), ("derivs", RDERIVS_DTYPE, MAXHARM)])
HIT_DTYPE = np.dtype([("power", "f4"), ("sigma", "f4"), ("numharm", "i4"), ("block", "i4"),
                      ("zind", "i4"), ("added", "i4"), ("rind", "i8"), ("r", "f8"), ("z", "f8")])
CAND_DTYPE = np.dtype([("power", "f4"), ("sigma", "f4"), ("numharm", "i4"), ("_pad", "i4"),
                       ("r", "f8"), ("z", "f8")])


def build(force=False):
    """Compile liboracle.so (and _ref/libprestoref.so when the reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "accel_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    elif os.path.isdir("/root/reference/src") and not os.path.exists(
            os.path.join(_HERE, "_ref", "libprestoref.so")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    d, i, ll, vp = C.c_double, C.c_int, C.c_longlong, C.c_void_p
    L.orc_fresnel.argtypes = [d, C.POINTER(d), C.POINTER(d)]
    L.orc_z_resp_halfwidth.argtypes = [d]; L.orc_z_resp_halfwidth.restype = i
    L.orc_next2_to_n.argtypes = [ll]; L.orc_next2_to_n.restype = ll
    L.orc_calc_required_z.argtypes = [d, d]; L.orc_calc_required_z.restype = i
    L.orc_calc_required_r.argtypes = [d, d]; L.orc_calc_required_r.restype = d
    L.orc_index_from_z.argtypes = [d, d]; L.orc_index_from_z.restype = i
    L.orc_calc_fftlen.argtypes = [i, i, i]; L.orc_calc_fftlen.restype = i
    L.orc_median.argtypes = [vp, i]; L.orc_median.restype = C.c_float
    L.orc_candidate_sigma.argtypes = [d, i, d]; L.orc_candidate_sigma.restype = d
    L.orc_power_for_sigma.argtypes = [d, i, d]; L.orc_power_for_sigma.restype = d
    L.orc_chi2_logp.argtypes = [d, i]; L.orc_chi2_logp.restype = d
    L.orc_equivalent_gaussian_sigma.argtypes = [d]; L.orc_equivalent_gaussian_sigma.restype = d
    L.orc_fft.argtypes = [vp, C.c_long, i]
    L.orc_gen_r_response.argtypes = [d, i, i, vp]
    L.orc_gen_z_response.argtypes = [d, i, d, i, vp]
    L.orc_place_complex_kernel.argtypes = [vp, i, vp, i]
    L.orc_init_kernel.argtypes = [i, i, vp, C.POINTER(i)]
    L.orc_obs_init.argtypes = [C.POINTER(Obs), vp, ll, ll, d, i, i, d, d, d, ll, i, i, i]
    L.orc_obs_init.restype = i
    L.orc_obs_free.argtypes = [C.POINTER(Obs)]
    L.orc_build_kernels.argtypes = [C.POINTER(Obs)]
    L.orc_fund_block.argtypes = [C.POINTER(Obs), d, vp, C.POINTER(i)]
    L.orc_fund_to_ffdotplane.argtypes = [C.POINTER(Obs), vp, i]
    L.orc_inmem_add_ffdotpows.argtypes = [C.POINTER(Obs), vp, i, i, i]
    L.orc_run.argtypes = [C.POINTER(Obs), vp, ll, C.POINTER(ll), vp, i, C.POINTER(i)]
    L.orc_run.restype = i
    L.orc_search_plane.argtypes = L.orc_run.argtypes
    L.orc_search_plane.restype = i
    L.orc_run_standard.argtypes = L.orc_run.argtypes
    L.orc_run_standard.restype = i
    pi = C.POINTER(i)
    L.orc_subharm_block.argtypes = [C.POINTER(Obs), i, i, d, d, vp, pi, pi, pi, pi, vp]
    L.orc_add_ffdotpows.argtypes = [C.POINTER(Obs), vp, vp, i, i, vp, i, i]
    L.orc_realfft_forward.argtypes = [vp, C.c_long]
    L.orc_deredden.argtypes = [vp, i]
    L.orc_zapbirds.argtypes = [vp, C.c_long, vp, vp, i]
    L.orc_rednoise.argtypes = [vp, C.c_long, i, i, d, d]
    L.orc_z_resp_halfwidth_high.argtypes = [d]; L.orc_z_resp_halfwidth_high.restype = i
    L.orc_rz_interp.argtypes = [vp, C.c_long, d, d, i, vp]
    L.orc_get_localpower3d.argtypes = [vp, C.c_long, d, d]; L.orc_get_localpower3d.restype = d
    L.orc_get_derivs3d.argtypes = [vp, C.c_long, d, d, d, vp]
    L.orc_max_rz.argtypes = [vp, C.c_long, i, i, d, d, i, C.POINTER(d), C.POINTER(d), vp, vp]
    L.orc_optimize_cand.argtypes = [C.POINTER(Obs), i, vp]
    L.orc_sort_and_eliminate.argtypes = [vp, i, i]; L.orc_sort_and_eliminate.restype = i
    L.orc_nice_output_1.argtypes = [C.c_char_p, d, d, i]; L.orc_nice_output_1.restype = i
    L.orc_nice_output_2.argtypes = [C.c_char_p, d, d, i]; L.orc_nice_output_2.restype = i
    L.orc_calc_props.argtypes = [vp, d, d, d, vp]; L.orc_calc_props.restype = i
    L.orc_write_accel_files.argtypes = [C.POINTER(Obs), vp, i, C.c_char_p, C.c_char_p, C.c_char_p]
    L.orc_write_accel_files.restype = i
    L.orc_fill_plane.argtypes = [C.POINTER(Obs)]
    L.orc_num_search_blocks.argtypes = [C.POINTER(Obs)]; L.orc_num_search_blocks.restype = i
    L.orc_num_prepop_blocks.argtypes = [C.POINTER(Obs)]; L.orc_num_prepop_blocks.restype = i
    _lib = L
    return L


def ref_lib():
    """The self-contained reference files built unmodified (None if not built)."""
    so = os.path.join(_HERE, "_ref", "libprestoref.so")
    if not os.path.exists(so):
        return None
    R = C.CDLL(so)
    d, i = C.c_double, C.c_int
    R.fresnl.argtypes = [d, C.POINTER(d), C.POINTER(d)]
    R.median.argtypes = [C.c_void_p, i]; R.median.restype = C.c_float
    pd, pi = C.POINTER(d), C.POINTER(i)
    R.cdfchi.argtypes = [pi, pd, pd, pd, pd, pi, pd]
    R.cdfnor.argtypes = [pi, pd, pd, pd, pd, pd, pi, pd]
    return R


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fft(x, isign):
    a = np.ascontiguousarray(x, dtype=np.complex64).copy()
    lib().orc_fft(_ptr(a), a.size, isign)
    return a


def gen_z_response(z, numkern, roffset=0.0, numbetween=2):
    out = np.empty(numkern, dtype=np.complex64)
    lib().orc_gen_z_response(roffset, numbetween, float(z), numkern, _ptr(out))
    return out


def init_kernel(z, fftlen):
    out = np.empty(fftlen, dtype=np.complex64)
    hw = C.c_int(0)
    lib().orc_init_kernel(int(z), fftlen, _ptr(out), C.byref(hw))
    return out, hw.value


def realfft(series):
    """Forward real FFT of a float32 series: complex64[n/2] in PRESTO's packed layout."""
    a = np.ascontiguousarray(series, dtype=np.float32).copy()
    lib().orc_realfft_forward(_ptr(a), a.size)
    return a.view(np.complex64)


def deredden(fft):
    a = np.ascontiguousarray(fft, dtype=np.complex64).copy()
    lib().orc_deredden(_ptr(a), a.size)
    return a


def zapbirds(fft, lobins, hibins):
    a = np.ascontiguousarray(fft, dtype=np.complex64).copy()
    lo = np.ascontiguousarray(lobins, dtype=np.float64)
    hi = np.ascontiguousarray(hibins, dtype=np.float64)
    lib().orc_zapbirds(_ptr(a), a.size, _ptr(lo), _ptr(hi), lo.size)
    return a


def rednoise(fft, T, startwidth=6, endwidth=100, endfreq=6.0):
    a = np.ascontiguousarray(fft, dtype=np.complex64).copy()
    lib().orc_rednoise(_ptr(a), a.size, startwidth, endwidth, float(endfreq), float(T))
    return a


def nice_output(error_digits, val, err):
    buf = C.create_string_buffer(128)
    fn = lib().orc_nice_output_1 if error_digits == 1 else lib().orc_nice_output_2
    rc = fn(buf, float(val), float(err), 0)
    return None if rc < 0 else buf.value.decode()


def median(a):
    b = np.ascontiguousarray(a, dtype=np.float32).copy()
    return float(lib().orc_median(_ptr(b), b.size))


class Search:
    """One accelsearch set-up on the CPU oracle (in-memory semantics)."""

    def __init__(self, fft, N, dt, zmax, numharm, sigma=2.0, flo=1.0, fhi=10000.0, lobin=0,
                 norm_type=0, photon=False, alloc_plane=True, nthreads=1):
        self.fft = np.ascontiguousarray(fft, dtype=np.complex64)
        self.obs = Obs()
        rc = lib().orc_obs_init(C.byref(self.obs), _ptr(self.fft), self.fft.size, int(N),
                                float(dt), int(zmax), int(numharm), float(sigma), float(flo),
                                float(fhi), int(lobin), int(norm_type), int(bool(photon)),
                                int(bool(alloc_plane)))
        if rc != 0:
            raise ValueError("orc_obs_init failed with %d" % rc)
        self.obs.nthreads = int(nthreads)
        self._kernels = False

    def close(self):
        lib().orc_obs_free(C.byref(self.obs))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- set-up values ----------------------------------------------------
    @property
    def numz(self): return self.obs.numz
    @property
    def fftlen(self): return self.obs.fftlen
    @property
    def rlo(self): return self.obs.rlo
    @property
    def highestbin(self): return self.obs.highestbin
    @property
    def numstages(self): return self.obs.numharmstages
    @property
    def powcut(self): return np.array(self.obs.powcut[:self.obs.numharmstages], dtype=np.float32)
    @property
    def numindep(self): return np.array(self.obs.numindep[:self.obs.numharmstages], dtype=np.int64)
    @property
    def rlen(self): return self.obs.rlen
    @property
    def num_blocks(self): return lib().orc_num_search_blocks(C.byref(self.obs))
    @property
    def num_prepop(self): return lib().orc_num_prepop_blocks(C.byref(self.obs))

    def build_kernels(self):
        if not self._kernels:
            lib().orc_build_kernels(C.byref(self.obs))
            self._kernels = True

    def kernel_bank(self):
        self.build_kernels()
        n = self.obs.numz * self.obs.fftlen
        buf = (C.c_float * (2 * n)).from_address(self.obs.kern)
        return np.frombuffer(buf, dtype=np.complex64).reshape(self.obs.numz, self.obs.fftlen).copy()

    def plane(self):
        """View (no copy) of the numz x rlen plane."""
        n = self.obs.numz * self.obs.rlen
        buf = (C.c_float * n).from_address(C.addressof(self.obs.ffdotplane.contents))
        return np.frombuffer(buf, dtype=np.float32).reshape(self.obs.numz, self.obs.rlen)

    def set_plane(self, arr):
        p = self.plane()
        p[...] = arr

    def fund_block(self, startr):
        self.build_kernels()
        out = np.empty((self.obs.numz, USELEN), dtype=np.float32)
        rlo = C.c_int(0)
        lib().orc_fund_block(C.byref(self.obs), float(startr), _ptr(out), C.byref(rlo))
        return out, rlo.value

    def fund_to_plane(self, powers, rlo):
        powers = np.ascontiguousarray(powers, dtype=np.float32)
        lib().orc_fund_to_ffdotplane(C.byref(self.obs), _ptr(powers), int(rlo))

    def inmem_add(self, powers, rlo, numharm, harmnum):
        assert powers.dtype == np.float32 and powers.flags.c_contiguous
        lib().orc_inmem_add_ffdotpows(C.byref(self.obs), _ptr(powers), int(rlo), numharm, harmnum)

    def fill_plane(self):
        self.build_kernels()
        lib().orc_fill_plane(C.byref(self.obs))

    def _collect(self, fn, max_hits, max_cands):
        hits = np.zeros(max_hits, dtype=HIT_DTYPE)
        cands = np.zeros(max_cands, dtype=CAND_DTYPE)
        nh, nc = C.c_longlong(0), C.c_int(0)
        fn(C.byref(self.obs), _ptr(hits), max_hits, C.byref(nh), _ptr(cands), max_cands,
           C.byref(nc))
        if nh.value > max_hits or nc.value > max_cands:
            raise RuntimeError("oracle buffers too small: %d hits, %d cands" % (nh.value, nc.value))
        return hits[:nh.value].copy(), cands[:nc.value].copy()

    def run(self, max_hits=4_000_000, max_cands=200_000):
        """Full main loop: returns (hits in visit order, final candidate list)."""
        self.build_kernels()
        return self._collect(lib().orc_run, max_hits, max_cands)

    def run_standard(self, max_hits=4_000_000, max_cands=200_000):
        """Main loop in standard (not in-memory) mode."""
        return self._collect(lib().orc_run_standard, max_hits, max_cands)

    def subharm_block(self, numharm, harmnum, fullrlo, fullrhi=None):
        """subharm_ffdot_plane for any harmonic: (powers[numzs, numrs], rlo, zlo, rinds)."""
        if fullrhi is None:
            fullrhi = fullrlo + USELEN * 0.5 - 0.5
        buf = np.zeros(self.obs.numz * (USELEN + 2), dtype=np.float32)
        rinds = np.zeros(USELEN, dtype=np.uint16)
        rlo, zlo, numrs, numzs = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        lib().orc_subharm_block(C.byref(self.obs), numharm, harmnum, float(fullrlo), float(fullrhi),
                                _ptr(buf), C.byref(rlo), C.byref(zlo), C.byref(numrs),
                                C.byref(numzs), _ptr(rinds))
        pw = buf[:numzs.value * numrs.value].reshape(numzs.value, numrs.value).copy()
        return pw, rlo.value, zlo.value, rinds

    def add_ffdotpows(self, fund, sub, sub_zlo, rinds, numharm, harmnum):
        assert fund.dtype == np.float32 and fund.flags.c_contiguous
        sub = np.ascontiguousarray(sub, dtype=np.float32)
        lib().orc_add_ffdotpows(C.byref(self.obs), _ptr(fund), _ptr(sub), int(sub_zlo), sub.shape[1],
                                _ptr(rinds), numharm, harmnum)

    def optimize(self, cands, polish=True, eliminate=True):
        """The steps after the search loop (src/accelsearch.c:225-254): sort by sigma,
        eliminate harmonics, optimise every survivor.  Returns OPTCAND_DTYPE records."""
        c = np.ascontiguousarray(cands, dtype=CAND_DTYPE).copy()
        n = lib().orc_sort_and_eliminate(_ptr(c), c.size, int(bool(eliminate)))
        out = np.zeros(n, dtype=OPTCAND_DTYPE)
        for k in ("power", "sigma", "numharm", "r", "z"):
            out[k] = c[k][:n]
        for j in range(n):
            one = out[j:j + 1]
            lib().orc_optimize_cand(C.byref(self.obs), int(bool(polish)), _ptr(one))
        return out

    def write_accel_files(self, optcands, accel_path, cand_path, inf_text=None):
        c = np.ascontiguousarray(optcands, dtype=OPTCAND_DTYPE).copy()
        rc = lib().orc_write_accel_files(C.byref(self.obs), _ptr(c), c.size, accel_path.encode(),
                                         cand_path.encode(),
                                         None if inf_text is None else inf_text.encode())
        if rc:
            raise ValueError("orc_write_accel_files failed")

    def search_plane(self, max_hits=4_000_000, max_cands=200_000):
        """Harmonic sum + threshold + list insertion on the plane as it stands."""
        return self._collect(lib().orc_search_plane, max_hits, max_cands)
