"""Usage: python3 scripts/pmc_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Turns the two rocprofv3 --pmc passes over `bench.py --timed-only` into the per-launch HBM byte
counts bench.py quotes as `roofline.traffic`.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 bytes and is doubled, WRITE_SIZE is exact.
"""
import collections
import csv
import json
import sys


def averages(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "accelk::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("accelk::", 1)[1].split("(", 1)[0]
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = averages(sys.argv[1], "FETCH_SIZE")
    write = averages(sys.argv[2], "WRITE_SIZE")
    per = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        per[k] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
                  "fetched_bytes": 2.0 * 1024.0 * f, "written_bytes": 1024.0 * w,
                  "hbm_bytes_per_launch": 2.0 * 1024.0 * f + 1024.0 * w}

    def total(prefixes):
        return sum(v["hbm_bytes_per_launch"] for k, v in per.items() if k.startswith(prefixes))

    prefilter = any(k.startswith("k_prefilter") for k in per)
    bounded = prefilter or any(k.startswith(("k_harmsum_b", "k_harmsum_bw", "k_harmsum_bl")) for k in per)
    out = {"workload": "cfg2 (2^23 pts, zmax 200, numharm 16), bench.py --timed-only",
           "prefilter": prefilter, "bounded": bounded,
           "harmsum_hbm_bytes_per_launch": total(("k_prefilter", "k_flagged_exact", "k_harmsum")),
           "correlate_hbm_bytes_per_launch": total(("k_correlate",)),
           "correction": "gfx950: FETCH_SIZE doubled (128-byte read requests tallied at 64 bytes), WRITE_SIZE exact; KiB",
           "kernels": per}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("prefilter", "bounded", "harmsum_hbm_bytes_per_launch",
                                           "correlate_hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
