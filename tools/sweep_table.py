"""Table of a tools/batch_sweep.sh run: per (configuration, batch) images/s, ms/step, whole-step fraction (algorithmic and by
bytes really moved) and the fraction relative to the batch-128 row of the same net.  A row more than 15 % below its batch-128
fraction is marked as a heuristic cliff."""
import json
import sys


def main(path):
    rows = []
    for ln in open(path):
        try:
            d = json.loads(ln)
        except Exception:
            continue
        if "sweep" in d:
            rows.append(d)
    ref = {}
    for d in rows:
        if d["sweep"]["batch"] == 128:
            ref[d["sweep"]["name"]] = d["roofline"]["whole_step"]["frac_algorithmic"]
    print("%-42s %5s %12s %9s %8s %8s %8s  %s" % ("configuration", "batch", "images/s", "ms/step", "frac_alg", "frac_act",
                                                 "vs b128", "dominant family (frac)"))
    for d in rows:
        s, r = d["sweep"], d["roofline"]
        w = r["whole_step"]
        base = ref.get(s["name"])
        rel = w["frac_algorithmic"] / base if base else float("nan")
        k = r.get("kernels", {})
        dom = max(k, key=lambda n: k[n]["ms_per_step"]) if k else "-"
        print("%-42s %5d %12.1f %9.4f %8.3f %8.3f %8.2f  %s (%.2f)%s" % (
            s["name"], s["batch"], d["value"], d["ms_per_step"], w["frac_algorithmic"], w["frac_actual"], rel, dom,
            k[dom]["frac"] if k else float("nan"), "   <-- CLIFF" if rel == rel and rel < 0.85 else ""))


if __name__ == "__main__":
    main(sys.argv[1])
