"""Copy one `scripts/profile_round.sh <tag>` result from gpurun_out/<tag> into profiles/<tag>:
json / stats files as they are, the raw --pmc counter rows filtered to the kernels the derived
json files are computed from (the full files list every torch set-up kernel as well)."""
import csv
import os
import shutil
import sys

KEEP = ("hist_accumulate_kernel", "hist_accumulate_multi_kernel", "prob3_events_kernel", "kde_pairs_kernel",
        "kde_lattice_kernel", "kde_h2l", "kde_hermite_coef", "kde_local_pilot", "barr_fold_multi_kernel")


def main(tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for name in sorted(os.listdir(src)):
        p = os.path.join(src, name)
        if not os.path.isfile(p) or name.endswith((".log", ".stderr")) and name != "c3_probe.log":
            continue
        if name.startswith("run_") and name.endswith(".json"):
            # stdout of a traced run: keep the JSON line only (RCCL / tool banners precede it)
            lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
            if lines:
                with open(os.path.join(dst, name), "w") as f:
                    f.write(lines[-1] + "\n")
            continue
        if name.startswith("pmc_") and name.endswith(".csv"):
            rows = list(csv.reader(open(p)))
            col = rows[0].index("Kernel_Name")
            kept = [rows[0]] + [r for r in rows[1:] if any(k in r[col] for k in KEEP)]
            with open(os.path.join(dst, name), "w", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_MINIMAL).writerows(kept)
            print("%-34s %5d of %5d rows" % (name, len(kept) - 1, len(rows) - 1))
        elif name == "kde_pmc_run.json":
            continue   # folded into kde_counter_check.json
        else:
            shutil.copy(p, os.path.join(dst, name))


if __name__ == "__main__":
    main(sys.argv[1])
