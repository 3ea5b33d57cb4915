#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof_*) into the small, tracked summaries
under profiles/:  <tag>_kernel_stats.csv (verbatim --stats table) and <tag>_pmc.json (per-kernel
means of each PMC counter + the derived HBM traffic per launch).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> doubled here."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


DENSE_SOURCES = ("ccqppy_amd/csrc/ccqp_gemv.hip", "ccqppy_amd/csrc/ccqp_bodies.h", "ccqppy_amd/csrc/ccqp_internal.h")


def source_hash(sources=DENSE_SOURCES):
    """sha1 over the sources a kernel is built from (default: the dense matvec, the same function as
    bench.gemv_source_hash; PROF_SOURCES = comma-separated paths names another kernel's)."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for rel in sources:
        with open(os.path.join(root, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def box():
    """The box the profile was taken on (this script runs right after the passes, on the same box)."""
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import box_id
        return box_id(0)
    except Exception as e:  # noqa: BLE001
        import socket
        return {"host": socket.gethostname(), "error": str(e)}


def main(tag, trace_dir, pmc_dirs, kernel_filter=None):
    kernel_filter = kernel_filter or os.environ.get("PROF_KERNEL_FILTER", "gemv")
    sources = tuple(os.environ["PROF_SOURCES"].split(",")) if os.environ.get("PROF_SOURCES") else DENSE_SOURCES
    where = box()
    os.makedirs("profiles", exist_ok=True)
    for f in glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, "profiles/%s_kernel_stats.csv" % tag)
        with open("profiles/%s_kernel_stats.box.json" % tag, "w") as fh:     # which box the table comes from
            json.dump({"box": where, "kernel_source_sha1": source_hash(sources)}, fh)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if kernel_filter in r["Kernel_Name"]]
            if os.environ.get("PROF_FULL_ONLY"):
                # a batched kernel whose launches carry a varying number of live problems: keep the launches that
                # carry the whole batch (duration >= 90 % of the longest), the ones `algorithmic_bytes_per_launch` means
                longest = collections.defaultdict(int)
                for r in rows:
                    longest[r["Kernel_Name"]] = max(longest[r["Kernel_Name"]], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                rows = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 0.9 * longest[r["Kernel_Name"]]]
            for r in rows:
                k = r["Kernel_Name"]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[k]["_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                           "grid_threads": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                           "lds": int(r["LDS_Block_Size"])}
    out = {}
    for k, c in agg.items():
        ncounters = max(len([n for n in c if n != "_ms"]), 1)
        e = {"launches_profiled_per_counter": len(c["_ms"]) // ncounters,
             "avg_ms_under_pmc": sum(c["_ms"]) / len(c["_ms"])}
        e.update(json.loads(os.environ.get("PROF_META", "{}")))
        e["kernel_source_sha1"] = source_hash(sources)      # bench.py reports this traffic only for the same sources
        e["box"] = where
        e.update(meta[k])
        for name, vals in c.items():
            if name != "_ms":
                e[name] = sum(vals) / len(vals)
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 * 2   # gfx950: x2 for wide streaming reads
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024
        if "hbm_read_bytes_per_launch" in e:
            e["hbm_traffic_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e.get("hbm_write_bytes_per_launch", 0.0)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        out[k] = e
    with open("profiles/%s_pmc.json" % tag, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
