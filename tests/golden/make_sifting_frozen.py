"""Freezes what presto_amd/sifting.py does on fixed synthetic inputs (tests/golden/sifting_frozen.json), so that
its behaviour cannot drift when its insides change.  These are outputs of this repository's own module at the
time of freezing -- NOT of the reference's lib/python/sifting.py, which cannot run here (Python 2 + a compiled
extension): they pin nothing against the reference.
    python tests/golden/make_sifting_frozen.py        (rewrites the JSON)
The input builders are imported by tests/test_sifting_cpu.py."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

T_OBS = 536.870912
DMS = [2.0 * k for k in range(14)]            # 0.00 .. 26.00


def synthetic_candidates(seed, n=260):
    """A list of Candidates over 14 DM trials with everything the cross-DM rules look at: clusters within r_err
    (tight ones, chains that only hold together through their strongest member, exact ties in sigma), integer
    and ratio harmonics of stronger candidates, peaks at low DM, hits with gaps in DM."""
    from presto_amd import sifting
    rng = np.random.default_rng(seed)
    cands = []
    num = {}

    def add(r, dm, sigma, nh=None, ipow=None):
        nh = int(rng.choice([1, 2, 4, 8, 16])) if nh is None else nh
        k = num[dm] = num.get(dm, 0) + 1
        ipow = float(np.round(sigma * 7.0 + rng.uniform(0, 3), 2)) if ipow is None else ipow
        c = sifting.Candidate(k, float(sigma), nh, ipow, float(np.round(ipow * 0.8, 2)), float(r), float(np.round(rng.normal(0, 20), 2)),
                              "%.2f" % dm, "beam_DM%.2f_ACCEL_200" % dm, T_OBS)
        c.harm_pows = np.full(nh, ipow / nh)
        c.snr = float(np.round(rng.uniform(3, 30), 2))
        c.hits = [(c.DM, c.snr, c.sigma)]
        cands.append(c)
        return c

    base = np.sort(rng.uniform(2000.0, 400000.0, 40))
    for b in base:
        kind = rng.integers(0, 6)
        dms = rng.choice(DMS, size=int(rng.integers(1, 9)), replace=False)
        peak = float(rng.uniform(7, 40))
        if kind == 0:                                   # a tight cluster around one r, one per DM
            for dm in dms:
                add(b + rng.uniform(-0.4, 0.4), dm, peak * rng.uniform(0.5, 1.0))
        elif kind == 1:                                 # a chain: neighbours within r_err, ends not
            for i, dm in enumerate(dms):
                add(b + 0.8 * i + rng.uniform(-0.05, 0.05), dm, peak * (1.0 - 0.1 * abs(i - len(dms) // 2)))
        elif kind == 2:                                 # exact ties in sigma (and sometimes in power)
            for dm in dms:
                add(b + rng.uniform(-0.5, 0.5), dm, float(np.round(peak, 1)), nh=4, ipow=100.0 if rng.random() < 0.5 else None)
        elif kind == 3:                                 # with harmonics elsewhere in the list
            for dm in dms:
                add(b + rng.uniform(-0.3, 0.3), dm, peak * rng.uniform(0.6, 1.0))
            for fac in rng.choice([2.0, 3.0, 0.5, 1.5, 2.5, 4.0 / 3.0, 0.25, 5.0, 0.2, 16.0], size=3, replace=False):
                add(b * fac + rng.uniform(-0.2, 0.2) * min(fac, 1.0), float(rng.choice(dms)), peak * rng.uniform(0.2, 1.2))
        elif kind == 4:                                 # strongest at the lowest DMs
            for dm in sorted(dms):
                add(b + rng.uniform(-0.3, 0.3), dm, peak / (1.0 + dm))
        else:                                           # hits two DMs apart only
            for dm in [4.0 * k for k in range(int(rng.integers(2, 5)))]:
                add(b + rng.uniform(-0.3, 0.3), dm, peak * rng.uniform(0.7, 1.0))
    while len(cands) < n:                               # unrelated singles
        add(rng.uniform(1000.0, 500000.0), float(rng.choice(DMS)), float(rng.uniform(2, 12)))
    order = rng.permutation(len(cands))
    return [cands[i] for i in order]


def ident(c):
    return "%s:%d" % (c.filename, c.candnum)


def describe(cl):
    """Everything observable about a Candlist after sifting steps."""
    return {
        "good": [[ident(c), c.sigma, c.r, [list(h) for h in c.hits]] for c in cl.cands],
        "duplicates": [[ident(c), c.note] for c in cl.duplicates],
        "bad": {k: [[ident(c), c.note] for c in v] for k, v in cl.badlists.items() if v},
    }


def run_case(seed):
    from presto_amd import sifting
    out = {}
    for step in ("duplicates", "harmonics", "dm", "sequence"):
        cl = sifting.Candlist(synthetic_candidates(seed), trackbad=True, trackdupes=True)
        if step == "duplicates":
            cl.remove_duplicate_candidates(verbosity=0)
        elif step == "harmonics":
            cl.remove_harmonics(verbosity=0)
        elif step == "dm":
            cl.remove_duplicate_candidates(verbosity=0)
            cl.remove_DM_problems(2, ["%.2f" % d for d in DMS], 2.0, verbosity=0)
        else:                                           # the order of bin/ACCEL_sift.py
            cl.default_rejection()
            cl.remove_duplicate_candidates(verbosity=0)
            cl.remove_DM_problems(2, ["%.2f" % d for d in DMS], 2.0, verbosity=0)
            cl.remove_harmonics(verbosity=0)
        out[step] = describe(cl)
    return out


def parsed_file(path):
    """What candlist_from_candfile makes of an ACCEL file, as plain numbers."""
    from presto_amd import sifting
    cl = sifting.candlist_from_candfile(path)
    return [[c.candnum, c.sigma, c.numharm, c.ipow_det, c.cpow, c.r, c.z, c.snr, [float(p) for p in c.harm_pows],
             [[float(a.real), float(a.imag)] for a in c.harm_amps]] for c in cl]


def big_accel_file(path, n=10020):
    """An ACCEL file with more than 9999 candidates: the candidate number overflows its four-character column."""
    import presto_amd as pa
    import presto_amd.api as api
    from presto_amd import synth
    N, dt = 1 << 22, 1e-4
    setup = pa.plan(N // 2, N, dt, 50, 8, fhi=1e9)
    rng = np.random.default_rng(77)
    c = np.zeros(n, dtype=api.OPTCAND_DTYPE)
    for i in range(n):
        nh = int(rng.choice([1, 2, 4]))
        hp = float(rng.uniform(5, 40))
        r = float(rng.uniform(1000, 2e6))
        c["r"][i], c["z"][i], c["numharm"][i], c["sigma"][i] = r, float(rng.normal(0, 10)), nh, 40.0 - i * 0.003
        c["power"][i] = hp * nh
        c["hirs"][i, :nh] = r * np.arange(1, nh + 1)
        c["hizs"][i, :nh] = c["z"][i] * np.arange(1, nh + 1)
        d = c["derivs"][i]
        d["locpow"][:nh] = 1.0
        d["pow"][:nh] = hp
        d["phs"][:nh] = rng.uniform(0.1, 6.0, nh)
        d["d2pow"][:nh] = -hp * 3.0
    api.write_accel_files(setup, c, path, path + ".cand", synth.inf_text("big_DM3.00", N, dt))
    return path


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()


def main():
    import tempfile
    # one case in full (a difference can be read off it), further ones as digests of the same description
    frozen = {"note": "outputs of presto_amd/sifting.py itself when frozen; they pin nothing against the reference",
              "cases": {"11": run_case(11)},
              "case_digests": {str(seed): {k: digest(v) for k, v in run_case(seed).items()} for seed in (11, 12, 13, 14, 15)}}
    small = parsed_file(os.path.join(HERE, "post_small_DM7.50_ACCEL_30"))
    frozen["post_small_parsed"] = small
    with tempfile.TemporaryDirectory() as tmp:
        big = parsed_file(big_accel_file(os.path.join(tmp, "big_DM3.00_ACCEL_50")))
    frozen["big_file"] = {"count": len(big), "digest": digest(big), "first": big[:2], "around_9999": big[9997:10002]}
    with open(os.path.join(HERE, "sifting_frozen.json"), "w") as f:
        json.dump(frozen, f, indent=0, sort_keys=True)
    print("frozen: %d cases, %d + %d parsed candidates" % (len(frozen["case_digests"]), len(small), len(big)))


if __name__ == "__main__":
    main()
