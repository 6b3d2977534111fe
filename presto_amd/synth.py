"""Deterministic synthetic inputs in PRESTO's on-disk formats.

A `.fft` file is the packed single-precision real-to-complex transform PRESTO's
`realfft` writes: N/2 complex float32 values, bin k = X[k] for k >= 1 and
bin 0 = (DC, Nyquist) (reference src/fastffts.c:251-256).  The `.inf` text is
the makedata flavour ("Telescope used = None (Artificial Data Set)"), which the
reference parser accepts with only N and dt (src/ioinf.c:121-175).
"""
import os

import numpy as np


def pulsar_series(N, dt, pulsars, seed, noise_sigma=1.0):
    """White noise plus pulse trains with a constant frequency derivative.

    pulsars: iterable of dicts {f (Hz, mean over the observation), z (Fourier
    bins of drift over T), amp, duty (FWHM fraction of a turn)}.
    Returns float64[N].
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal(N) * noise_sigma
    T = N * dt
    chunk = 1 << 20
    for p in pulsars:
        r_mean = p["f"] * T
        z = p.get("z", 0.0)
        r_start = r_mean - 0.5 * z
        kappa = np.log(2.0) / (2.0 * np.sin(np.pi * p.get("duty", 0.05) / 2.0) ** 2)
        for lo in range(0, N, chunk):
            u = (np.arange(lo, min(lo + chunk, N), dtype=np.float64) + 0.5) / N
            phase = r_start * u + 0.5 * z * u * u
            x[lo:lo + u.size] += p["amp"] * np.exp(kappa * (np.cos(2 * np.pi * phase) - 1.0))
    return x


def packed_fft(x):
    """float64[N] -> complex64[N/2] in PRESTO's packed layout."""
    N = x.size
    X = np.fft.rfft(x)
    out = np.empty(N // 2, dtype=np.complex64)
    out[1:] = X[1:N // 2].astype(np.complex64)
    out[0] = np.complex64(complex(X[0].real, X[N // 2].real))
    return out


def make_fft(N, dt, pulsars, seed):
    """Synthetic spectrum with unit mean noise power.  Returns complex64[N/2]."""
    x = pulsar_series(N, dt, pulsars, seed)
    # E|X_k|^2 = N * sigma^2 for white noise; scale to unit mean power
    return packed_fft(x / np.sqrt(N))


def noise_fft(numbins, seed, lines=()):
    """Spectrum drawn directly in the Fourier domain: unit-mean-power complex
    Gaussian noise plus optional coherent lines [(r, z, amp)] injected through
    their exact drifting-tone response.  Much cheaper than a time series."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.standard_normal((numbins, 2)).astype(np.float32) * np.float32(np.sqrt(0.5))
    out = a.view(np.complex64).reshape(numbins).copy()
    for (r, z, amp) in lines:
        lo = max(int(r - abs(z) - 40), 1)
        hi = min(int(r + abs(z) + 40), numbins - 1)
        k = np.arange(lo, hi)
        out[lo:hi] += (amp * chirp_response(k - r, z)).astype(np.complex64)
    return out


def chirp_response(dr, z, nquad=4096):
    """Fourier response at offsets dr (bins from the mean frequency) of a unit
    tone drifting linearly by z bins: int_0^1 exp(2 pi i (-(dr) u + z (u^2 - u)/2)) du,
    by midpoint quadrature (vectorised)."""
    u = (np.arange(nquad) + 0.5) / nquad
    ph = -np.outer(np.atleast_1d(dr), u) + 0.5 * z * (u * u - u)
    return np.exp(2j * np.pi * ph).mean(axis=1)


def write_fft(path_root, spectrum, N, dt):
    """Write <root>.fft and <root>.inf."""
    spectrum.astype(np.complex64).tofile(path_root + ".fft")
    with open(path_root + ".inf", "w") as f:
        f.write(inf_text(os.path.basename(path_root), N, dt))


def inf_text(name, N, dt):
    rows = [
        (" Data file name without suffix", name),
        (" Telescope used", "None (Artificial Data Set)"),
        (" Number of bins in the time series", "%d" % N),
        (" Width of each time series bin (sec)", "%.15g" % dt),
        (" Any breaks in the data? (1 yes, 0 no)", "0"),
        (" Data analyzed by", "presto_amd"),
    ]
    txt = "".join("%-40s =  %s\n" % r for r in rows)
    return txt + " Any additional notes:\n    synthetic data\n"


# The workloads BASELINE.json names, as (N, dt, zmax, numharm, pulsars).
CONFIGS = {
    "cfg1": dict(N=1 << 20, dt=1e-4, zmax=0, numharm=8, seed=1,
                 pulsars=[dict(f=123.456789, z=0.0, amp=0.25, duty=0.03)]),
    "cfg2": dict(N=1 << 23, dt=64e-6, zmax=200, numharm=16, seed=2,
                 pulsars=[dict(f=211.3, z=37.4, amp=0.08, duty=0.04),
                          dict(f=3.17, z=0.0, amp=0.05, duty=0.06),
                          dict(f=1417.9, z=-150.2, amp=0.12, duty=0.05)]),
    "cfg5": dict(N=1 << 24, dt=64e-6, zmax=1200, numharm=32, seed=5,
                 pulsars=[dict(f=311.7, z=1100.0, amp=0.4, duty=0.05),
                          dict(f=57.3, z=-420.6, amp=0.15, duty=0.05)]),
}


def config_fft(name, N=None, seed=None):
    """Spectrum for a named workload (optionally at a reduced N for CPU-sized runs)."""
    c = dict(CONFIGS[name])
    if N is not None:
        c["N"] = N
    if seed is not None:
        c["seed"] = seed
    return make_fft(c["N"], c["dt"], c["pulsars"], c["seed"]), c


def chirp_fft(N, r, z):
    """Packed spectrum of the noiseless unit-amplitude drifting tone cos(2 pi (r0 u + z u^2 / 2)),
    u = t/N, r0 = r - z/2: mean Fourier frequency r bins, drift z bins over the series (the
    signal of the reference's python/ffdot_example.py:8-21).  A filter matched in (r, z)
    recovers the power N^2/4."""
    u = np.arange(N, dtype=np.float64) / N
    return packed_fft(np.cos(2.0 * np.pi * (u * (u * (0.5 * z) + (r - 0.5 * z)))))
