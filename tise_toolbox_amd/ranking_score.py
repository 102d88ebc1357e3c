#!/usr/bin/env python3
"""Ranking score table: the step after the metric scripts.

Mirror of the reference's `ranking_scores/ranking_score.py` (whole file, :1-80): every method's nine aspect
metrics are read from `methods/<METHOD>.json`, each metric is ranked across methods (FID, O-FID and CA are
lower-is-better: sign flipped before ranking, :33-36; rank 1 = worst, rank n = best, via `np.argsort` down the
method axis, :37-47), and the ranking score is the sum of the six aspect scores (:50-62)

    RS = mean(rank IS*, rank FID) + rank RP + mean(rank SOA-C, rank SOA-I) + mean(rank O-IS, rank O-FID)
         + rank CA + rank PA

The table (metrics + RS per method) is rendered with pandas + tabulate `psql` exactly as the reference does
(:72-74), written to `results/coco_benchmark_results.txt` and printed (:77-79).  Run without arguments from a
directory holding `methods/` and `results/` it behaves like the reference script.

Added for the build: `--collect NAME` writes `methods/NAME.json` from the result files this package's metric
scripts save (`fid_score.py --saved_file`, `inception_score.py --saved_file`, the O-IS / O-FID forms), with
`--set KEY=VALUE` for metrics produced elsewhere (SOA, CA, PA, RP) -- the wiring the reference leaves to hand
editing (README.md:437-441).  Host bookkeeping only: nothing here touches the GPU.
"""
import argparse
import json
import os
import re
from collections import OrderedDict

import numpy as np

METRICS = ["IS*", "FID", "RP", "SOA-C", "SOA-I", "O-IS", "O-FID", "CA", "PA"]      # ranking_score.py:11
LOWER_IS_BETTER = ("FID", "O-FID", "CA")                                            # ranking_score.py:34-36


def list_methods(methods_dir="methods"):
    """Method names in directory order, with the reference's file filter (ranking_score.py:14):
    `name.split(".")[1] == "json"`, method = `name.split(".")[0]`."""
    return [f.split(".")[0] for f in os.listdir(methods_dir) if f.split(".")[1] == "json"]


def load_scores(methods, methods_dir="methods"):
    """OrderedDict method -> list of the nine metric values as float (ranking_score.py:17-23)."""
    scores = OrderedDict()
    for method in methods:
        with open(os.path.join(methods_dir, f"{method}.json"), "r") as f:
            method_scores = json.load(f)
        scores[method] = [float(method_scores[metric]) for metric in METRICS]
    return scores


def metric_ranks(score_matrix):
    """(n_methods, n_metrics) ranks, 1 = worst .. n = best, ties broken by `np.argsort` as in the reference."""
    s = np.array(score_matrix, dtype=np.float64)
    for name in LOWER_IS_BETTER:
        s[:, METRICS.index(name)] = -s[:, METRICS.index(name)]
    order = np.argsort(s, 0)                       # row r of column m: index of the method with the r-th lowest value
    ranks = np.zeros(s.shape)
    n = s.shape[0]
    for metric_idx in range(s.shape[1]):
        ranks[order[:, metric_idx], metric_idx] = np.arange(1, n + 1)
    return ranks


def ranking_score(ranks_row):
    """Sum of the six aspect scores of one method (ranking_score.py:52-62)."""
    r = dict(zip(METRICS, ranks_row))
    aspects = [np.mean([r["IS*"], r["FID"]]), r["RP"], np.mean([r["SOA-C"], r["SOA-I"]]),
               np.mean([r["O-IS"], r["O-FID"]]), r["CA"], r["PA"]]
    return float(np.sum(aspects))


def ranking_table(scores):
    """scores: OrderedDict method -> nine floats.  Returns (methods, array (n, 10) = metrics + RS)."""
    methods = list(scores)
    mat = np.array([scores[m] for m in methods], dtype=np.float64)
    ranks = metric_ranks(mat)
    rs = np.array([ranking_score(ranks[i]) for i in range(len(methods))])
    return methods, np.concatenate([mat, rs[:, None]], 1)


def format_table(methods, table):
    """The reference's rendering (ranking_score.py:72-74)."""
    import pandas as pd
    from tabulate import tabulate
    df = pd.DataFrame(table, columns=METRICS + ["RS"])
    df.insert(loc=0, column="Method", value=methods)
    return tabulate(df, headers="keys", tablefmt="psql", showindex=False)


def compute(methods_dir="methods", methods=None):
    methods = list_methods(methods_dir) if methods is None else list(methods)
    names, table = ranking_table(load_scores(methods, methods_dir))
    return names, table, format_table(names, table)


# ---- wiring to the metric scripts' result files --------------------------------------------------------------
_RESULT_PATTERNS = {
    "FID": r"^FID:\s*([-+0-9.eE]+)",                                   # fid_score.py --saved_file
    "O-FID": r"^O-FID:\s*([-+0-9.eE]+)",                               # fid_score.py --label O-FID
    "IS*": r"\[Inception Score\]\s*mean:\s*([-+0-9.eE]+)",             # inception_score.py (coco form)
    "O-IS": r"^O-IS:\s*([-+0-9.eE]+)",                                 # object_centric_inception_score.py
    "RP": r"^R-precision:\s*([-+0-9.eE]+)",                            # RP_coco.py:85-90 form
}


def parse_result_file(metric, path):
    """Value of `metric` from a result file written by the corresponding script."""
    text = open(path).read()
    if "[synthetic weights" in text:                       # weights.SYNTHETIC_TAG: a plumbing run, not a score
        raise ValueError(f"{path}: produced with --synthetic-weights (seeded stand-in parameters); not a real {metric}")
    m = re.search(_RESULT_PATTERNS[metric], text, re.M)
    if not m:
        raise ValueError(f"{path}: no {metric} result line")
    return float(m.group(1))


def collect(name, files, extra, methods_dir="methods"):
    """Write methods/<name>.json from result files (metric -> path) and literal values (metric -> value)."""
    out = OrderedDict()
    for metric in METRICS:
        if metric in files:
            out[metric] = parse_result_file(metric, files[metric])
        elif metric in extra:
            out[metric] = float(extra[metric])
        else:
            raise ValueError(f"metric {metric} missing: give --set '{metric}=VALUE' or its result file")
    os.makedirs(methods_dir, exist_ok=True)
    path = os.path.join(methods_dir, f"{name}.json")
    with open(path, "w") as f:
        json.dump(out, f)
    return path


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--methods-dir", default="methods")
    ap.add_argument("--results-file", default=os.path.join("results", "coco_benchmark_results.txt"))
    ap.add_argument("--collect", metavar="NAME", default=None, help="write methods/NAME.json from result files")
    ap.add_argument("--fid", default=None); ap.add_argument("--ofid", default=None)
    ap.add_argument("--is", dest="is_file", default=None); ap.add_argument("--ois", default=None)
    ap.add_argument("--rp", default=None)
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE")
    args = ap.parse_args(argv)
    if args.collect:
        files = {k: v for k, v in (("FID", args.fid), ("O-FID", args.ofid), ("IS*", args.is_file), ("O-IS", args.ois),
                                   ("RP", args.rp)) if v}
        extra = dict(kv.split("=", 1) for kv in args.set)
        print(collect(args.collect, files, extra, args.methods_dir))
        return
    _, _, text = compute(args.methods_dir)
    with open(args.results_file, "w") as f:
        f.write(text)
    print(text)


if __name__ == "__main__":
    main()
