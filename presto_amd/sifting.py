"""Cross-DM sifting of acceleration-search candidates.

Host-side consumer of the per-trial result files (or of the candidate records
gathered over RCCL by presto_amd.batch): preliminary rejections, duplicate
removal across DM trials, harmonic removal, DM-consistency checks and the
candidate list report.  Same names, arguments, thresholds and results as the
reference's module (lib/python/sifting.py); plotting is not part of it.

Module-level thresholds can be overwritten by the importing script, as there
(lib/python/sifting.py:13-38).
"""
import copy
import os
import re
import sys

import numpy as np

# -- thresholds (reference lib/python/sifting.py:18-38) --------------------------
r_err = 1.1              # bins: closer than this is the same candidate
long_period = 15.0       # s
short_period = 0.0005    # s
sigma_threshold = 6.0
c_pow_threshold = 100.0
harm_pow_cutoff = 8.0
known_birds_p = []       # (ms, one-sided width)
known_birds_f = []       # (Hz, one-sided width)

fund_re = re.compile(r"^\d")
harms_re = re.compile(r"^[ ]\d")
DM_re = re.compile(r"DM(\d+\.\d{2})")

_BADLISTS = ("knownbirds", "longperiod", "shortperiod", "threshold", "harmpowcutoff",
             "rogueharmpow", "harmonic", "dmproblem")
_HEADER = ("#" + "file:candnum".center(66) + "DM".center(9) + "SNR".center(8) + "sigma".center(8) +
           "numharm".center(9) + "ipow".center(9) + "cpow".center(9) + "P(ms)".center(14) +
           "r".center(12) + "z".center(8) + "numhits".center(9) + "\n")


def candidate_sigma(power, numsum, numtrials):
    """Equivalent Gaussian sigma of a summed power (the library's host routine;
    reference src/characteristics.c:548-568)."""
    from . import api
    return api.candidate_sigma(power, numsum, numtrials)


def parse_power(text):
    """'12.3(4)' or '1.2(3)x10^5' -> float (reference :132-140)."""
    power = float(text.split("(")[0])
    if "^" in text:
        try:
            expon = float(text.split("^")[1])
        except ValueError:
            expon = 5
        power *= 10.0 ** expon
    return power


def _by_sigma(c):          # decreasing sigma, ties by decreasing detected power (:143-148)
    return (-c.sigma, -c.ipow_det)


class Candidate(object):
    """One fundamental of one ACCEL file (reference :167-205)."""

    def __init__(self, candnum, sigma, numharm, ipow, cpow, bin, z, DMstr, filename, T):
        self.path, self.filename = os.path.split(filename)
        self.candnum = candnum
        self.sigma = sigma
        self.numharm = numharm
        self.ipow_det = ipow
        self.cpow = cpow
        self.r = bin
        self.f = bin / T
        self.z = z
        self.T = T
        self.p = 1.0 / self.f
        self.DMstr = DMstr
        self.DM = float(DMstr)
        self.harm_pows = None
        self.harm_amps = None
        self.snr = 0.0
        self.hits = []
        self.note = ""

    def add_as_hit(self, other):
        self.hits.extend(other.hits)

    def __str__(self):
        cand = self.filename + ":" + repr(self.candnum)
        return "%-65s   %7.2f  %6.2f  %6.2f  %s   %7.1f  %7.1f  %12.6f  %10.2f  %8.2f " % (
            cand, self.DM, self.snr, self.sigma, "%2d".center(7) % self.numharm, self.ipow_det,
            self.cpow, self.p * 1000, self.r, self.z)

    def harms_to_snr(self):
        amps = np.asarray(self.harm_pows) - 1.0       # remove the mean noise power
        amps[amps < 0.0] = 0.0
        self.snr = np.sum(np.sqrt(amps))


class Candlist(object):
    """Good candidates plus, optionally, what was rejected and why (reference :208-1097)."""

    def __init__(self, cands=None, trackbad=False, trackdupes=False):
        self.cands = [] if cands is None else cands
        self.trackbad = trackbad
        self.trackdupes = trackdupes
        self.badlists = dict((k, []) for k in _BADLISTS)
        self.duplicates = []

    def __iter__(self):
        return iter(self.cands)

    def __getitem__(self, key):
        return self.cands[key]

    def __delitem__(self, key):
        del self.cands[key]

    def __len__(self):
        return len(self.cands)

    def get_numcands(self):
        return len(self)

    def sort(self, *args, **kwargs):
        self.cands.sort(*args, **kwargs)

    def mark_as_bad(self, icand, badlistname):
        cand = self.cands.pop(icand)
        if self.trackbad:
            self.badlists.setdefault(badlistname, []).append(cand)

    def mark_as_duplicate(self, icand):
        cand = self.cands.pop(icand)
        if self.trackdupes:
            self.duplicates.append(cand)

    def get_all_goodcands(self):
        return self.cands + self.duplicates

    def get_all_badcands(self):
        out = []
        for key in self.badlists.keys():
            out += self.badlists[key]
        return out

    def get_all_cands(self):
        return self.get_all_goodcands() + self.get_all_badcands()

    # -- preliminary rejections (:536-730) ---------------------------------------
    def _reject(self, test, badlist):
        for ii in reversed(range(len(self.cands))):
            note = test(self.cands[ii])
            if note:
                self.cands[ii].note = note
                self.mark_as_bad(ii, badlist)

    def reject_longperiod(self, long_period=None):
        lim = globals()["long_period"] if long_period is None else long_period
        self._reject(lambda c: c.p > lim and "Period is too long (%g ms > %g ms)" % (c.p * 1000, lim * 1000),
                     "longperiod")

    def reject_shortperiod(self, short_period=None):
        lim = globals()["short_period"] if short_period is None else short_period
        self._reject(lambda c: c.p < lim and "Period is too short (%g ms < %g ms)" % (c.p * 1000, lim * 1000),
                     "shortperiod")

    def reject_knownbirds(self, known_birds_f=[], known_birds_p=[]):
        if known_birds_f is None:
            known_birds_f = globals()["known_birds_f"]
        if known_birds_p is None:
            known_birds_p = globals()["known_birds_p"]

        def test(c):
            for bird, err in known_birds_f:
                if np.fabs(c.f - bird) < err:
                    return ("Freq (%.2f Hz) is within %g Hz of a known birdie centred at %.2f Hz"
                            % (c.f, err, bird))
            for bird, err in known_birds_p:
                if np.fabs(c.p * 1000.0 - bird) < err:
                    return ("Period (%.2f ms) is within %g ms of a known birdie centred at %.2f ms"
                            % (c.f * 1000, err, bird))
            return None
        self._reject(test, "knownbirds")

    def reject_threshold(self, sigma_threshold=None, c_pow_threshold=None):
        sig = globals()["sigma_threshold"] if sigma_threshold is None else sigma_threshold
        cpw = globals()["c_pow_threshold"] if c_pow_threshold is None else c_pow_threshold

        def test(c):
            if c.numharm == 1:
                if c.sigma < sig and c.cpow < cpw:
                    return ("Only 1 harmonic and both sigma (%g < %g) and coherent power (%g < %g) are "
                            "too low." % (c.sigma, sig, c.cpow, cpw))
            elif c.sigma < sig:
                return "%d harmonics and sigma (%g < %g) is too low." % (c.numharm, c.sigma, sig)
            return None
        self._reject(test, "threshold")

    def reject_harmpowcutoff(self, harm_pow_cutoff=None):
        cut = globals()["harm_pow_cutoff"] if harm_pow_cutoff is None else harm_pow_cutoff
        self._reject(lambda c: np.max(c.harm_pows) < cut and "All harmonics have power < %g" % cut,
                     "harmpowcutoff")

    def reject_rogueharmpow(self):
        def test(c):
            maxharm = int(np.argmax(c.harm_pows))
            maxpow = c.harm_pows[maxharm]
            ranked = np.sort(c.harm_pows)
            if ((c.numharm >= 8 and maxharm > 4 and maxpow > 2 * ranked[-2]) or
                    (c.numharm >= 4 and maxharm > 2 and maxpow > 3 * ranked[-2])):
                return "High-numbered harmonic (%d) has too much power" % maxharm
            return None
        self._reject(test, "rogueharmpow")

    def default_rejection(self):
        self.reject_longperiod()
        self.reject_shortperiod()
        self.reject_knownbirds()
        self.reject_threshold()
        self.reject_harmpowcutoff()
        self.reject_rogueharmpow()

    # -- across DM trials (:732-967) -----------------------------------------------
    def remove_duplicate_candidates(self, verbosity=1):
        """Keep the most significant of the candidates within r_err bins of each other
        and give it the others' hits."""
        if verbosity >= 1:
            print("  Sorting the %d candidates by frequency..." % self.get_numcands())
        self.cands.sort(key=lambda c: c.r)
        if verbosity >= 1:
            print("  Searching for dupes...")
        ii = 0
        while ii < len(self.cands):
            jj = ii + 1
            if jj < len(self.cands) and np.fabs(self.cands[ii].r - self.cands[jj].r) < r_err:
                jj += 1
                while jj < len(self.cands) and np.fabs(self.cands[ii].r - self.cands[jj].r) < r_err:
                    jj += 1
                best = sorted(self.cands[ii:jj], key=_by_sigma)[0]
                bestindex = self.cands.index(best)
                for k in reversed(range(ii, jj)):
                    if k == bestindex:
                        continue
                    match = self.cands[k]
                    best.add_as_hit(match)
                    match.note = "This candidate is a duplicate of %s:%d" % (best.filename, best.candnum)
                    self.mark_as_duplicate(k)
                    if verbosity >= 2:
                        print("Removing %s:%d (index: %d)" % (match.filename, match.candnum, k))
                        print("    %s" % match.note)
                # the best one now sits at ii; look again, more may match it
            else:
                ii += 1
        if verbosity >= 1:
            print("Found %d candidates.\n" % self.get_numcands())
        self.cands.sort(key=_by_sigma)

    def remove_harmonics(self, verbosity=1):
        """Drop the less significant of harmonically related candidates."""
        numremoved = 0
        self.cands.sort(key=_by_sigma)
        if len(self.cands) < 2:
            return
        f_err = r_err / self.cands[0].T
        if verbosity >= 1:
            print("\nSearching for duplicate harmonics...")
        ratios = list(zip([3.0, 5.0, 2.0, 4.0, 5.0, 3.0, 5.0, 2.0, 3.0, 4.0],
                          [2.0, 2.0, 3.0, 3.0, 3.0, 4.0, 4.0, 5.0, 5.0, 5.0]))
        ii = 0
        while True:
            fund = self.cands[ii]
            jj = len(self.cands) - 1
            while True:
                harm = self.cands[jj]
                harmstr = None
                for factor in np.arange(1.0, 17.0):
                    if np.fabs(fund.f - harm.f * factor) < f_err * factor:
                        harmstr = "1/%dth" % factor
                    elif np.fabs(fund.f - harm.f / factor) < f_err / factor:
                        harmstr = ("%dnd" if factor == 2.0 else "%dth") % factor
                    if harmstr:
                        break
                for numer, denom in ratios:
                    factor = numer / denom
                    if np.fabs(fund.f - harm.f * factor) < f_err * factor:
                        harmstr = "%d/%dth" % (denom, numer)
                        break
                if harmstr:
                    harm.note = ("This candidate (P=%.4f s, DM=%.2f) is a harmonic (%s) of %s:%d "
                                 "(P=%.4f s, DM=%.2f)." % (harm.p, harm.DM, harmstr, fund.filename,
                                                           fund.candnum, fund.p, fund.DM))
                    numremoved += 1
                    self.mark_as_bad(jj, "harmonic")
                    if verbosity >= 2:
                        print("Removing %s:%d (index: %d)" % (harm.filename, harm.candnum, jj))
                        print("    %s" % harm.note)
                jj -= 1
                if jj == ii:
                    break
            ii += 1
            if ii >= len(self.cands) - 1:
                break
        if verbosity >= 1:
            print("Removed a total of %d harmonics.\n" % numremoved)

    def remove_DM_problems(self, numdms, dmlist, low_DM_cutoff, verbosity=1):
        """Drop candidates with fewer than numdms hits, whose strongest hit is at or below
        low_DM_cutoff, or whose hits are never at neighbouring DMs."""
        dms = np.unique([float(dm) for dm in dmlist])
        dmdict = dict(zip(["%.2f" % dm for dm in dms], range(len(dms))))
        counts = dict(toofew=0, toolow=0, gaps=0)
        self.cands.sort(key=_by_sigma)
        for ii in reversed(range(len(self.cands))):
            c = self.cands[ii]
            why = None
            if len(c.hits) < numdms:
                why = "toofew"
                c.note = ("Candidate has only %d DM hits. This is less than minimum for 'good' cands "
                          "(%d hits)" % (len(c.hits), numdms))
            else:
                imax = int(np.argmax(np.array([hit[2] for hit in c.hits])))
                hitdm, hitsnr, hitsigma = c.hits[imax]
                if float(hitdm) <= low_DM_cutoff:
                    why = "toolow"
                    c.note = ("Hit with max sigma (%g) has dm (%.2f) <= low DM cutoff (%.2f) "
                              % (hitsigma, hitdm, low_DM_cutoff))
                elif len(c.hits) > 1:
                    c.hits.sort(key=lambda h: float(h[0]))
                    idx = np.asarray([dmdict["%.2f" % h[0]] for h in c.hits])
                    if min(idx[1:] - idx[:-1]) > 1:
                        why = "gaps"
                        c.note = "DM list of hits has gaps (i.e. consecutive DMs don't have hits)."
            if why:
                counts[why] += 1
                self.mark_as_bad(ii, "dmproblem")
                if verbosity >= 2:
                    print("Removing %s:%d (index: %d)" % (c.filename, c.candnum, ii))
                    print("    %s" % c.note)
        if verbosity >= 1:
            print("Removed %d candidates with DM problems.\n" % sum(counts.values()))
        if verbosity >= 2:
            print("  # with too few hits:", counts["toofew"])
            print("  # with peak SNR too low:", counts["toolow"])
            print("  # with gaps in DM hits:", counts["gaps"])

    # -- reports (:968-1097) ----------------------------------------------------------
    def print_cand_summary(self, summaryfilenm=None):
        lines = ["   Candlist contains %d 'good' candidates\n" % len(self.cands)]
        for label, n in (("Known RFI rejects:          ", len(self.badlists["knownbirds"])),
                         ("Short period rejects:       ", len(self.badlists["shortperiod"])),
                         ("Long period rejects:        ", len(self.badlists["longperiod"])),
                         ("Missed threshold:           ", len(self.badlists["threshold"])),
                         ("No good harmonics:          ", len(self.badlists["harmpowcutoff"])),
                         ("One bad harmonic:           ", len(self.badlists["rogueharmpow"])),
                         ("Duplicate candidates:       ", len(self.duplicates)),
                         ("Harmonic candidates:        ", len(self.badlists["harmonic"])),
                         ("Candidates with DM problems:", len(self.badlists["dmproblem"]))):
            lines.append("      # %s %d\n" % (label, n))
        self._emit(summaryfilenm, "".join(lines))

    def write_cand_report(self, reportfilenm=None):
        text = _HEADER
        for bad in self.get_all_badcands():
            text += "%s (%d)\n" % (str(bad), len(bad.hits))
            text += "    Note: %s\n\n" % bad.note
        self._emit(reportfilenm, text)

    def to_file(self, candfilenm=None):
        text = _HEADER
        for good in self.cands:
            text += "%s (%d)\n" % (str(good), len(good.hits))
            if len(good.hits) > 1:
                good.hits.sort(key=lambda h: float(h[0]))
                for hit in good.hits:
                    text += "  DM=%6.2f SNR=%5.2f Sigma=%5.2f   " % tuple(hit) + int(hit[2] / 3.0) * "*" + "\n"
        self._emit(candfilenm, text)

    @staticmethod
    def _emit(dest, text):
        if dest is None:
            sys.stdout.write(text)
        elif dest in (sys.stdout, sys.stderr):
            dest.write(text)
        else:
            with open(dest, "w") as f:
                f.write(text)

    def extend(self, other):
        self.cands.extend(other.cands)
        self.duplicates.extend(other.duplicates)
        for key in other.badlists:
            self.badlists.setdefault(key, []).extend(other.badlists[key])

    def __add__(self, other):
        out = copy.deepcopy(self)
        out.extend(other)
        return out


# -- the old function-style API (:46-104) ----------------------------------------------
def remove_duplicate_candidates(candlist, *args, **kwargs):
    if not isinstance(candlist, Candlist):
        candlist = Candlist(candlist)
    candlist.remove_duplicate_candidates(*args, **kwargs)
    return candlist


def remove_DM_problems(candlist, *args, **kwargs):
    if not isinstance(candlist, Candlist):
        candlist = Candlist(candlist)
    candlist.remove_DM_problems(*args, **kwargs)
    return candlist


def remove_harmonics(candlist, *args, **kwargs):
    if not isinstance(candlist, Candlist):
        candlist = Candlist(candlist)
    candlist.remove_harmonics(*args, **kwargs)
    return candlist


def write_candlist(candlist, *args, **kwargs):
    candlist.to_file(*args, **kwargs)


# -- input ------------------------------------------------------------------------------
def _finish(cand):
    """Statistics the reference derives once all harmonic powers are known (:1153-1195)."""
    cand.harms_to_snr()
    opt_ipow = cand.harm_pows[0] if cand.numharm == 1 else sum(cand.harm_pows)
    cand.sigma = candidate_sigma(opt_ipow, cand.numharm, 1)
    cand.hits = [(cand.DM, cand.snr, cand.sigma)]
    cand.ipow_det = opt_ipow


def candlist_from_candfile(filename, trackbad=False, trackdupes=False):
    """Parse one ACCEL text file (reference :1099-1200)."""
    with open(filename, "r") as f:
        lines = f.readlines()
    numsamp = dt = None
    for line in lines:
        if line.startswith(" Number of bins in the time series"):
            numsamp = int(line.split()[-1])
        if line.startswith(" Width of each time series bin (sec)"):
            dt = float(line.split()[-1])
    tobs = numsamp * dt
    DMstr = DM_re.search(filename).groups()[0]
    cands, candnums = [], []
    pending, pending_harm = None, 0
    last_candnum = last_good = 0
    for line in lines:
        if fund_re.match(line):
            s = line.split()
            candnum = int(s[0])
            if cands and len(s[0]) == 4 and cands[-1].candnum >= 9999:
                candnum = last_candnum + 1
            bin_ = float(s[7].split("(")[0])
            z = float(s[9].split("(")[0])
            cands.append(Candidate(candnum, float(s[1]), int(s[4]), float(s[2]), float(s[3]), bin_, z,
                                   DMstr, filename, tobs))
            candnums.append(candnum)
            last_candnum = candnum
            continue
        if harms_re.match(line):
            s = line.split()
            candnum = int(s[0])
            if len(s[0]) == 4 and last_good >= 9999:
                candnum = last_good + 1
            if candnum in candnums:
                cand = cands[candnums.index(candnum)]
                cand.harm_pows = np.zeros(cand.numharm, dtype=np.float64)
                cand.harm_amps = np.zeros(cand.numharm, dtype=np.complex64)
                power = parse_power(s[3])
                phase = float(s[9].split("(")[0])
                cand.harm_pows[0] = power
                cand.harm_amps[0] = np.sqrt(power) * np.exp(phase * 1.0j)
                if cand.numharm > 1:
                    pending, pending_harm = cand, 1
                else:
                    last_good = candnum
                    pending = None
                    _finish(cand)
            continue
        if pending is not None:
            s = line.split()
            power = parse_power(s[2])
            phase = float(s[8].split("(")[0])
            pending.harm_pows[pending_harm] = power
            pending.harm_amps[pending_harm] = np.sqrt(power) * np.exp(phase * 1.0j)
            pending_harm += 1
            if pending_harm == pending.numharm:
                _finish(pending)
                last_good = pending.candnum
                pending = None
    return Candlist(cands, trackbad=trackbad, trackdupes=trackdupes)


def candlist_from_optcands(optcands, setup, DMstr, filename, nph=0.0, trackbad=False,
                           trackdupes=False):
    """The same Candlist straight from optimised candidates in memory (the records
    presto_amd.api.AccelSearchGPU.optimize returns), skipping the text file.  Values
    are taken at the precision the file would print them with, so both routes give
    identical lists."""
    from . import api
    tobs = setup.T
    cands = []
    for i, q in enumerate(api.sift_records(optcands, nph)):
        nh = int(q["numharm"])
        cand = Candidate(i + 1, float(q["sigma"]), nh, float(q["ipow"]), float(q["cpow"]), float(q["r"]),
                         float(q["z"]), DMstr, filename, tobs)
        pows = np.array(q["harm_pow"][:nh])
        cand.harm_pows = pows
        cand.harm_amps = (np.sqrt(pows) * np.exp(q["harm_phs"][:nh] * 1.0j)).astype(np.complex64)
        _finish(cand)
        cands.append(cand)
    return Candlist(cands, trackbad=trackbad, trackdupes=trackdupes)


def read_candidates(filenms, prelim_reject=True, track=False):
    """Candidates of many ACCEL files in one list (reference :1203-1229)."""
    candlist = Candlist(trackbad=track, trackdupes=track)
    if not filenms:
        print("Error:  There are no candidate files to read!")
        return candlist
    print("\nReading candidates from %d files...." % len(filenms))
    for ii, filenm in enumerate(filenms):
        cur = candlist_from_candfile(filenm, trackbad=track, trackdupes=track)
        if prelim_reject:
            cur.default_rejection()
        candlist.extend(cur)
    print("\nDone")
    return candlist
