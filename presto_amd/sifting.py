"""Cross-DM sifting of acceleration-search candidates.

Host-side consumer of the per-trial result files (or of the candidate records
gathered over RCCL by presto_amd.batch): preliminary rejections, duplicate
removal across DM trials, harmonic removal, DM-consistency checks and the
candidate list report.  Names, arguments, thresholds, notes and file formats are
those of the reference's module (lib/python/sifting.py), so that scripts written
against it keep working; the cross-DM steps and the ACCEL-file parser are written
afresh on columns and index sets (one stable sort by r with bisected windows for the
duplicates, one broadcast comparison per fundamental for the harmonics, a verdict
function for the DM tests, block-wise reading of the harmonics table).  Plotting is
not part of it.  The reference module cannot run here (Python 2 + a compiled
extension): tests are known answers plus outputs frozen from this module itself.

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

    # -- across DM trials (rules of the reference's :732-967) ------------------------------
    # The three steps below work on columns (r, f, sigma, power as arrays) and index sets rather than by
    # popping list elements one at a time; what they keep, drop and note is what the reference's rules say.
    def _columns(self):
        c = self.cands
        return (np.array([x.r for x in c], dtype=np.float64), np.array([x.f for x in c], dtype=np.float64),
                np.array([x.sigma for x in c], dtype=np.float64), np.array([x.ipow_det for x in c], dtype=np.float64))

    def remove_duplicate_candidates(self, verbosity=1):
        """Candidates within r_err bins of each other are one signal seen in several trials: the most
        significant stays and collects the others' hits.  A run is measured from its first member in
        frequency order; when a stronger member takes over, the run is measured again from that one, so a
        chain can grow past r_err from where it began."""
        n = len(self.cands)
        if verbosity >= 1:
            print("  Sorting the %d candidates by frequency..." % n)
        r, _, sig, ipw = self._columns()
        order = np.argsort(r, kind="stable")          # equal r: in the order given
        rs = r[order]
        if verbosity >= 1:
            print("  Searching for dupes...")
        cands = self.cands
        kept = []
        i = 0
        while i < n:
            head = int(order[i])                      # the candidate the run is measured from
            j = i + 1                                 # first position (in frequency order) not yet looked at
            while True:
                # sorted by r, so whoever lies within r_err of `head` from j on is one stretch: a bisection
                # finds about where it ends, the comparison itself (as the rule states it) says exactly
                end = max(int(np.searchsorted(rs, r[head] + r_err, side="left")), j)
                while end > j and not np.fabs(r[head] - rs[end - 1]) < r_err:
                    end -= 1
                while end < n and np.fabs(r[head] - rs[end]) < r_err:
                    end += 1
                if end == j:
                    break
                members = [head] + [int(k) for k in order[j:end]]
                # largest sigma, then largest detected power, then first in the run
                best = min(range(len(members)), key=lambda k: (-sig[members[k]], -ipw[members[k]]))
                strongest = cands[members[best]]
                for k in reversed(range(len(members))):
                    if k == best:
                        continue
                    match = cands[members[k]]
                    strongest.add_as_hit(match)
                    match.note = "This candidate is a duplicate of %s:%d" % (strongest.filename, strongest.candnum)
                    if self.trackdupes:
                        self.duplicates.append(match)
                    if verbosity >= 2:
                        print("Removing %s:%d (index: %d)" % (match.filename, match.candnum, len(kept) + k))
                        print("    %s" % match.note)
                head, j = members[best], end
            kept.append(head)
            i = j
        self.cands = [cands[k] for k in kept]
        if verbosity >= 1:
            print("Found %d candidates.\n" % self.get_numcands())
        self.cands.sort(key=_by_sigma)

    # harmonic relations looked for besides the integer ones, as (numerator, denominator) of fund.f / harm.f
    _RATIOS = ((3, 2), (5, 2), (2, 3), (4, 3), (5, 3), (3, 4), (5, 4), (2, 5), (3, 5), (4, 5))

    def remove_harmonics(self, verbosity=1):
        """Going down the list by significance, whatever lies at an integer multiple or fraction (up to 16)
        or at one of the small ratios of a stronger candidate's frequency goes."""
        self.cands.sort(key=_by_sigma)
        n = len(self.cands)
        if n < 2:
            return
        cands = self.cands
        _, f, _, _ = self._columns()
        f_err = r_err / cands[0].T
        if verbosity >= 1:
            print("\nSearching for duplicate harmonics...")
        ks = np.arange(1.0, 17.0)
        ratio = np.array([float(a) / float(b) for a, b in self._RATIOS])
        alive = np.ones(n, dtype=bool)
        nalive, processed, fund, numremoved = n, 0, 0, 0
        while True:
            later = fund + 1 + np.flatnonzero(alive[fund + 1:])
            fh = f[later][:, None]
            # one row per weaker candidate, one column per factor: is harm.f * k (then harm.f / k) the fundamental's frequency?
            below = np.fabs(f[fund] - fh * ks) < f_err * ks
            above = np.fabs(f[fund] - fh / ks) < f_err / ks
            whole = below | above
            frac = np.fabs(f[fund] - fh * ratio) < f_err * ratio
            for row in np.flatnonzero(whole.any(axis=1) | frac.any(axis=1))[::-1]:     # weakest first
                jj = int(later[row])
                if frac[row].any():                   # a ratio, if one fits, is what the note names
                    numer, denom = self._RATIOS[int(np.argmax(frac[row]))]
                    harmstr = "%d/%dth" % (denom, numer)
                else:
                    k = int(np.argmax(whole[row]))    # the smallest factor that fits
                    harmstr = "1/%dth" % (k + 1) if below[row, k] else ("%dnd" if k == 1 else "%dth") % (k + 1)
                harm, top = cands[jj], cands[fund]
                harm.note = ("This candidate (P=%.4f s, DM=%.2f) is a harmonic (%s) of %s:%d "
                             "(P=%.4f s, DM=%.2f)." % (harm.p, harm.DM, harmstr, top.filename, top.candnum, top.p, top.DM))
                alive[jj] = False
                nalive -= 1
                numremoved += 1
                if self.trackbad:
                    self.badlists.setdefault("harmonic", []).append(harm)
                if verbosity >= 2:
                    print("Removing %s:%d (index: %d)" % (harm.filename, harm.candnum, processed + 1 + int(row)))
                    print("    %s" % harm.note)
            processed += 1
            if processed >= nalive - 1:               # the last one left has nobody below it
                break
            fund = fund + 1 + int(np.argmax(alive[fund + 1:]))
        self.cands = [c for c, ok in zip(cands, alive) if ok]
        if verbosity >= 1:
            print("Removed a total of %d harmonics.\n" % numremoved)

    @staticmethod
    def _dm_verdict(c, numdms, place, low_DM_cutoff):
        """Why a candidate's hits over the DM trials do not look like a pulsar's, or None: (reason, note)."""
        if len(c.hits) < numdms:
            return "toofew", ("Candidate has only %d DM hits. This is less than minimum for 'good' cands "
                              "(%d hits)" % (len(c.hits), numdms))
        sigmas = np.array([h[2] for h in c.hits])
        hitdm, _, hitsigma = c.hits[int(np.argmax(sigmas))]
        if float(hitdm) <= low_DM_cutoff:
            return "toolow", ("Hit with max sigma (%g) has dm (%.2f) <= low DM cutoff (%.2f) "
                              % (hitsigma, hitdm, low_DM_cutoff))
        if len(c.hits) > 1:
            c.hits.sort(key=lambda h: float(h[0]))
            where = np.array([place["%.2f" % h[0]] for h in c.hits])
            if np.diff(where).min() > 1:
                return "gaps", "DM list of hits has gaps (i.e. consecutive DMs don't have hits)."
        return None

    def remove_DM_problems(self, numdms, dmlist, low_DM_cutoff, verbosity=1):
        """Drop candidates with fewer than numdms hits, whose strongest hit is at or below
        low_DM_cutoff, or whose hits are never at neighbouring DMs."""
        dms = np.unique([float(dm) for dm in dmlist])
        place = dict(("%.2f" % dm, i) for i, dm in enumerate(dms))     # a trial's position among the DMs searched
        self.cands.sort(key=_by_sigma)
        verdicts = [self._dm_verdict(c, numdms, place, low_DM_cutoff) for c in self.cands]
        counts = dict(toofew=0, toolow=0, gaps=0)
        for ii in reversed(range(len(self.cands))):   # (the rejects are listed weakest first)
            if verdicts[ii] is None:
                continue
            c = self.cands[ii]
            why, c.note = verdicts[ii]
            counts[why] += 1
            if self.trackbad:
                self.badlists.setdefault("dmproblem", []).append(c)
            if verbosity >= 2:
                print("Removing %s:%d (index: %d)" % (c.filename, c.candnum, ii))
                print("    %s" % c.note)
        self.cands = [c for c, v in zip(self.cands, verdicts) if v is None]
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


_INF_N_RE = re.compile(r"^ Number of bins in the time series.*?(\S+)[ \t]*$", re.M)
_INF_DT_RE = re.compile(r"^ Width of each time series bin \(sec\).*?(\S+)[ \t]*$", re.M)


def _value(field):
    """'123.45(6)' -> 123.45"""
    return float(field.split("(")[0])


def candlist_from_candfile(filename, trackbad=False, trackdupes=False):
    """One ACCEL text file as a Candlist (what the reference's parser, :1099-1200, makes of it).

    The file is three parts: a table with one row per fundamental (rows begin with the candidate number in the
    first column), a table of harmonics in blocks of `numharm` rows (the first row of a block begins with a blank
    and the candidate number, the others with blanks only), and the .inf text, which holds N and dt.  Candidate
    numbers are printed in four characters: from 10 000 on the column overflows and the numbers are counted on
    from the last one read instead."""
    with open(filename, "r") as f:
        text = f.read()
    tobs = int(_INF_N_RE.findall(text)[-1]) * float(_INF_DT_RE.findall(text)[-1])
    DMstr = DM_re.search(filename).groups()[0]
    lines = text.split("\n")
    # the fundamentals
    cands, by_number = [], {}
    for fields in (ln.split() for ln in lines if fund_re.match(ln)):
        number = int(fields[0])
        if cands and len(fields[0]) == 4 and cands[-1].candnum >= 9999:
            number = cands[-1].candnum + 1
        cand = Candidate(number, float(fields[1]), int(fields[4]), float(fields[2]), float(fields[3]), _value(fields[7]),
                         _value(fields[9]), DMstr, filename, tobs)
        cands.append(cand)
        by_number.setdefault(number, cand)
    # the harmonics: a block per candidate, as many rows as it has harmonics
    last_done, at = 0, 0
    while at < len(lines):
        if not harms_re.match(lines[at]) or fund_re.match(lines[at]):
            at += 1
            continue
        fields = lines[at].split()
        number = int(fields[0])
        if len(fields[0]) == 4 and last_done >= 9999:
            number = last_done + 1
        cand = by_number.get(number)
        if cand is None:
            at += 1
            continue
        # the block's rows as (power field, phase field): the first row carries the number in front
        rows = [(fields[3], fields[9])] + [(p[2], p[8]) for p in (ln.split() for ln in lines[at + 1:at + cand.numharm])]
        if len(rows) == cand.numharm:
            power = np.array([parse_power(p) for p, _ in rows], dtype=np.float64)
            phase = np.array([_value(q) for _, q in rows], dtype=np.float64)
            cand.harm_pows = power
            cand.harm_amps = (np.sqrt(power) * np.exp(phase * 1.0j)).astype(np.complex64)
            _finish(cand)
            last_done = number
        at += max(cand.numharm, 1)
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
