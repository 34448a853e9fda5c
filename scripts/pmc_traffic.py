"""HBM traffic per kernel launch from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE and
WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").

usage: python scripts/pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json [WORKLOAD]
(WORKLOAD, e.g. cifar / resnet128_dstep: the table is merged into OUT.json["workloads"][WORKLOAD],
which is what bench.py reads for `roofline.traffic` and the D-step leg's `traffic`)

Correction (MI355X_MICROARCH.md "HBM"): on gfx950 FETCH_SIZE tallies the 128-B read requests of a
wide coalesced stream at 64 B, i.e. reports half the bytes -> hbm_bytes = (2 * FETCH_SIZE +
WRITE_SIZE) * 1024 (both counters are in KiB).  WRITE_SIZE is uncalibrated (taken as is).
Kernel names are folded into the same families bench.py's HIP-event brackets use."""
import collections, csv, glob, json, os, re, sys

FAMILIES = [
    (r"halo_conv_kernel", "halo_conv_kernel<*>"),
    (r"hconv_kernel<128", "hconv_kernel<128, *>"),
    (r"hconv(_rw)?_kernel", "hconv_kernel<64, *>"),
    (r"hup_kernel", "hconv_kernel<64, *>"),            # bracketed under the 64-channel-tile family
    (r"(thin_conv|small_linear)_kernel", "gconv_kernel<...>"),
    (r"hwgrad_kernel", "hwgrad_kernel<*>"),
    (r"sconv_kernel", "sconv_kernel<*>"),
    (r"swgrad_kernel", "swgrad_kernel<*>"),
    (r"wstem_fwd_kernel", "stem_fwd_kernel<*>"),
    (r"wstem_wgrad_kernel", "stem_wgrad_kernel<*>"),
    (r"fast_conv(_sk)?_kernel<128, 192", "fast_conv_kernel<128, 192, *>"),
    (r"fast_conv(_sk)?_kernel<128, 128", "fast_conv_kernel<128, 128, *>"),
    (r"fast_conv(_sk)?_kernel<64, 128", "fast_conv_kernel<64, 128, *>"),
    (r"fast_conv(_sk)?_kernel<128, 64", "fast_conv_kernel<128, 64, *>"),
    (r"fast_conv(_sk)?_kernel<128, 32", "fast_conv_kernel<128, 32, *>"),
    (r"stem_fwd_kernel", "stem_fwd_kernel<*>"),
    (r"gconv_kernel", "gconv_kernel<...>"),
    (r"halo_wgrad_kernel", "halo_wgrad_kernel<*>"),
    (r"fast_wgrad_kernel<128", "fast_wgrad_kernel<128, *>"),
    (r"fast_wgrad_kernel<64", "fast_wgrad_kernel<64, *>"),
    (r"stem_wgrad_kernel", "stem_wgrad_kernel<*>"),
    (r"gwgrad_kernel", "gwgrad_kernel<...>"),
]


def family(name):
    for pat, fam in FAMILIES:
        if re.search(pat, name):
            return fam
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def collect(d, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            fam = family(row["Kernel_Name"])
            tot[fam] += float(row["Counter_Value"])
            n[fam].add(row["Dispatch_Id"])
    return {k: (tot[k], len(n[k])) for k in tot}


def main():
    fetch, write, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else None
    fs, ws = collect(fetch, "FETCH_SIZE"), collect(write, "WRITE_SIZE")
    res = {}
    for fam in sorted(set(fs) | set(ws)):
        f_tot, f_n = fs.get(fam, (0.0, 0))
        w_tot, w_n = ws.get(fam, (0.0, 0))
        f_avg = f_tot / f_n if f_n else 0.0
        w_avg = w_tot / w_n if w_n else 0.0
        res[fam] = {"launches_fetch_pass": f_n, "launches_write_pass": w_n,
                    "FETCH_SIZE_KiB_raw": round(f_avg, 2), "WRITE_SIZE_KiB": round(w_avg, 2),
                    "hbm_bytes_per_launch": int((2.0 * f_avg + w_avg) * 1024)}
    table = {"correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE "
                           "counts 128-B requests at 64 B; MI355X_MICROARCH.md HBM section)",
             "workload": "eager launches of the same kernels as the captured step",
             "families": res}
    if workload:
        doc = {"workloads": {}}
        if os.path.exists(out):
            doc = json.load(open(out))
            doc.setdefault("workloads", {})
        doc["workloads"][workload] = table
        json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
    else:
        json.dump(table, open(out, "w"), indent=1, sort_keys=True)
    for fam, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:14]:
        print("%-36s %8.2f MB/launch (%d launches)" % (fam, v["hbm_bytes_per_launch"] / 1e6,
                                                       v["launches_fetch_pass"]))


if __name__ == "__main__":
    main()
