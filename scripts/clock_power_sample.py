"""Shader clock and socket power while one kernel runs back to back for a few seconds: plane generation, then the
harmonic sum, then nothing.  Samples come from the amdgpu driver's sysfs files of the card (what rocm-smi prints),
read every 50 ms by a thread while the main thread keeps the GPU busy.
python scripts/clock_power_sample.py   (one GPU)"""
import glob
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import presto_amd as pa                      # noqa: E402
from presto_amd import synth                 # noqa: E402


def find_card():
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            cards.append(dev)
    return cards


def read_sclk(dev):
    """MHz of the level marked with '*' in pp_dpm_sclk."""
    try:
        for line in open(os.path.join(dev, "pp_dpm_sclk")):
            if "*" in line:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except Exception:
        pass
    return float("nan")


def read_power(dev):
    """Watts: hwmon power1_average (or power1_input), microwatts in the file."""
    for name in ("power1_average", "power1_input"):
        for path in glob.glob(os.path.join(dev, "hwmon", "hwmon*", name)):
            try:
                return float(open(path).read()) * 1e-6
            except Exception:
                pass
    return float("nan")


def busy_percent(dev):
    try:
        return float(open(os.path.join(dev, "gpu_busy_percent")).read())
    except Exception:
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, devs):
        super().__init__(daemon=True)
        self.devs, self.rows, self.stop = devs, [], False

    def run(self):
        while not self.stop:
            self.rows.append([(read_sclk(d), read_power(d), busy_percent(d)) for d in self.devs])
            time.sleep(0.05)


def main():
    devs = find_card()
    print("cards with a clock table: %s" % devs, flush=True)
    numbins, dt, zmax, nh = 1 << 22, 64e-6, 200, 16
    g = pa.AccelSearchGPU(0)
    setup = g.configure(numbins, 2 * numbins, dt, zmax, nh, fhi=1e9)
    g.load_fft(synth.make_fft(2 * numbins, dt, dict(synth.CONFIGS["cfg2"])["pulsars"], seed=1000))
    g.set_profiling(True)
    g.run()
    g.timings()

    def phase(name, fn, seconds=3.0):
        s = Sampler(devs)
        s.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
        g.synchronize()
        wall = time.perf_counter() - t0
        s.stop = True
        s.join()
        tm = g.timings()
        rows = np.array(s.rows[len(s.rows) // 4:])          # the first quarter: ramping up
        # the busiest card is the one this process runs on
        k = int(np.nanargmax(np.nanmean(rows[:, :, 1], axis=0))) if rows.size else 0
        sc, pw, bz = rows[:, k, 0], rows[:, k, 1], rows[:, k, 2]
        kern = {kk: v["total_ms"] / max(v["launches"], 1) for kk, v in tm.items() if v["launches"]}
        print("%-22s %5d calls in %.2f s (%.3f ms each); kernel ms %s; card %d: sclk %.0f MHz (min %.0f, max %.0f), "
              "power %.0f W (max %.0f), busy %.0f %%  [%d samples]" % (
                  name, n, wall, 1e3 * wall / max(n, 1), {a: round(b, 3) for a, b in kern.items()}, k, np.nanmean(sc),
                  np.nanmin(sc), np.nanmax(sc), np.nanmean(pw), np.nanmax(pw), np.nanmean(bz), len(sc)), flush=True)

    phase("idle", lambda: time.sleep(0.05), 1.5)
    phase("plane generation", g.fill_plane)
    phase("harmonic sum", lambda: g.search_plane(max_hits=1 << 20))
    phase("whole search", lambda: g.run(max_hits=1 << 20))
    g.set_tuning("hs_bound", 3)
    phase("generation, no maxima", g.fill_plane)
    g.set_tuning("hs_bound", 1)
    phase("idle", lambda: time.sleep(0.05), 1.5)
    g.close()


if __name__ == "__main__":
    main()
