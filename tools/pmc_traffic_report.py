"""Fold the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into HBM bytes per launch per kernel class.

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (FETCH_SIZE/WRITE_SIZE are reported in KiB; the factor 2 is the gfx950
correction of MI355X_MICROARCH.md §HBM).  The calibration rows (a 1 GiB torch copy: 2^30 B read, 2^30 B written) are
printed next to the result so the correction can be judged on this very run."""
import collections, csv, glob, json, sys

d, batch = sys.argv[1], int(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else "darcy"
CLASSES = [("conv7x7", "conv"), ("conv1x1", "conv"), ("conv_igemm", "conv"), ("conv3x3_stream", "conv"), ("conv3x3_split", "conv"), ("conv3x3_rs", "conv"), ("conv_wgrad", "conv"), ("lap_", "attn"), ("wgrad_reduce", "conv_aux"), ("reduce_multi", "conv_aux"),
           ("pack_", "conv_aux"), ("clip_adam", "optimizer"), ("sqsum", "optimizer"),
           ("colsum", "conv_aux"), ("gn_", "norm"), ("layernorm", "norm"), ("la_", "attn"), ("mid_attn", "attn"),
           ("darcy", "darcy"), ("qsample", "darcy"), ("mech_", "mechanics"), ("bilinear", "mechanics"), ("psample", "sampler")]


def cls(name):
    for pat, c in CLASSES:
        if pat in name:
            return c
    return "torch_elementwise_and_copies" if ("elementwise" in name or "copy" in name.lower()) and "pidm" not in name else "other"


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
big = collections.defaultdict(dict)
outside = collections.defaultdict(float)
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{d}/{counter}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        # the two steps = everything between the first and the last kernel of the library: model initialisation, the synthetic
        # batch and the 1 GiB calibration copy (torch kernels before the first step) are NOT step traffic (rounds 1-3 counted them:
        # ~1.7 GB per "step" in the torch class)
        ids = [int(r["Dispatch_Id"]) for r in rows if "pidm::" in r["Kernel_Name"]]
        lo, hi = (min(ids), max(ids)) if ids else (0, 1 << 62)
        for r in rows:
            c = cls(r["Kernel_Name"])
            v = float(r["Counter_Value"])
            if c == "torch_elementwise_and_copies":
                big[counter][r["Dispatch_Id"]] = max(big[counter].get(r["Dispatch_Id"], 0.0), v)
            if not (lo <= int(r["Dispatch_Id"]) <= hi):
                outside[counter] += v
                continue
            tot[c][counter] += v
            cnt[c][counter] += 1
res = {"batch": batch, "workload": workload, "round": "round 6",
       "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches; 2 steps (1 warm-up + 1 timed); kernels before the first / "
               "after the last library kernel (initialisation, calibration copy) are excluded",
       "excluded_outside_the_steps_bytes": (2 * outside["FETCH_SIZE"] + outside["WRITE_SIZE"]) * 1024}
for c in tot:
    n = max(cnt[c]["FETCH_SIZE"], cnt[c]["WRITE_SIZE"], 1)
    fk, wk = tot[c]["FETCH_SIZE"], tot[c]["WRITE_SIZE"]
    res[c] = {"launches": n, "fetch_KiB": fk, "write_KiB": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024 / n,
              "hbm_bytes_per_step": (2 * fk + wk) * 1024 / 2}
cal = {k: max(v.values()) if v else None for k, v in big.items()}
res["calibration_1GiB_copy"] = {"FETCH_SIZE_KiB_max_dispatch": cal.get("FETCH_SIZE"), "WRITE_SIZE_KiB_max_dispatch": cal.get("WRITE_SIZE"),
                                "expected_KiB_each": 2 ** 20}
json.dump(res, open(f"{d}/traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
