"""profiles/<tag>_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-graph`:

    python tools/make_traffic.py <tag> <pmc_FETCH_SIZE dir> <pmc_WRITE_SIZE dir> [kernel substring = conv_dma_kernel<3]

Records the mean HBM-side bytes per launch of the dominant kernel family (2 * FETCH_SIZE + WRITE_SIZE: gfx950 tallies 128-byte read
requests as 64 B, MI355X_MICROARCH.md HBM section) together with a hash of the kernel sources; bench.py refuses the number
(`roofline.traffic: null`, a line on stderr) when the sources it runs no longer hash to it.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_hash  # noqa: E402


def per_kernel(root: str, counter: str, filt: str) -> dict:
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if filt in k and row["Counter_Name"] == counter:
                    acc[k].append(float(row["Counter_Value"]))
    return acc


def main() -> None:
    tag, fdir, wdir = sys.argv[1:4]
    filt = sys.argv[4] if len(sys.argv) > 4 else "conv_dma_kernel<3"
    fe, wr = per_kernel(fdir, "FETCH_SIZE", filt), per_kernel(wdir, "WRITE_SIZE", filt)
    nf, nw = sum(len(v) for v in fe.values()), sum(len(v) for v in wr.values())
    if not nf or not nw:
        raise SystemExit(f"no {filt} dispatches in the counter files")
    fetch_kb = sum(sum(v) for v in fe.values()) / nf
    write_kb = sum(sum(v) for v in wr.values()) / nw
    out = {"family": "conv3x3_dma", "kernels": sorted(k[:100] for k in fe), "dispatches": nf, "fetch_size_kb_mean": round(fetch_kb, 1),
           "write_size_kb_mean": round(write_kb, 1), "traffic_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024), "csrc_sha256": csrc_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 2 --warmup 1 --no-graph`; "
                   "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B); mean over the family's dispatches"}
    path = os.path.join(ROOT, "profiles", f"{tag}_traffic.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
