"""From the per-kernel PMC summaries of tools/profile_round.sh (pmc_FETCH_SIZE.csv / pmc_WRITE_SIZE.csv): the HBM traffic per launch of the
dominant kernel (gate/up projection) as the JSON bench.py replays into `roofline.traffic`.

    python tools/pmc_gateup_json.py <dir with pmc_FETCH_SIZE.csv> <out.json> [note]
"""
import csv
import json
import os
import sys


def rows(path, counter):
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        if r["counter"] == counter and "lsk_gemm_kernel<1, 2," in r["kernel"]:
            out[r["kernel"]] = (int(r["launches"]), float(r["mean"]))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    fetch = rows(os.path.join(src, "pmc_FETCH_SIZE.csv"), "FETCH_SIZE")
    write = rows(os.path.join(src, "pmc_WRITE_SIZE.csv"), "WRITE_SIZE")
    assert fetch, "no gate/up rows in pmc_FETCH_SIZE.csv"
    n = sum(c for c, _ in fetch.values())
    mean_kb = sum(c * m for c, m in fetch.values()) / n
    one = [m for k, (c, m) in fetch.items() if "<1, 2, 1," in k]
    multi = [m for k, (c, m) in fetch.items() if "<1, 2, 1," not in k]
    out = {
        "kernel": "lsk_gemm_kernel<PRO_RMS,EPI_SWIGLU> (MB=1 draft passes and MB=8 verify passes)",
        "hbm_read_bytes_per_launch": int(mean_kb * 1024 * 2),
        "hbm_read_bytes_per_launch_1row": int(one[0] * 1024 * 2) if one else None,
        "hbm_read_bytes_per_launch_multirow": int(multi[0] * 1024 * 2) if multi else None,
        "write_kb_per_launch_raw_1row": next((m for k, (c, m) in write.items() if "<1, 2, 1," in k), None),
        "launches": n,
        "method": "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), "
                  "bench.py --steps 1 --warmup 0 --max-steps 48; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide "
                  "coalesced stream); KB = 1024 B" + ("; " + note if note else ""),
        "algorithmic_bytes": 2 * 2 * 11008 * 4096,
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
