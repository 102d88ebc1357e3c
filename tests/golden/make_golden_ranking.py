#!/usr/bin/env python3
"""Golden vectors for the ranking-score table.

Runs the reference's own script (ranking_scores/ranking_score.py, executed BY PATH from a scratch working
directory that holds a copy of its methods/*.json data files, so nothing is written under /root/reference) and
stores what it printed together with the os.listdir order it saw, which fixes the row order.  Also stores the
reference repository's committed result table (results/coco_benchmark_results.txt) and the method JSONs
themselves: data fixtures, no source.

    python tests/golden/make_golden_ranking.py          (needs /root/reference; run in the build container)
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ranking_scores"


def main():
    out_dir = os.path.join(HERE, "ranking")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(out_dir, "methods"), os.path.join(tmp, "methods"))
        os.makedirs(os.path.join(tmp, "results"))
        order = os.listdir(os.path.join(tmp, "methods"))
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        r = subprocess.run([sys.executable, os.path.join(REF, "ranking_score.py")], cwd=tmp, env=env, check=True,
                           capture_output=True, text=True)
        saved = open(os.path.join(tmp, "results", "coco_benchmark_results.txt")).read()
    assert r.stdout.rstrip("\n") == saved, "reference prints what it saves"
    json.dump({"listdir_order": order, "table": saved}, open(os.path.join(out_dir, "reference_run.json"), "w"), indent=1)
    # a second case with ties and a different method count: perturbed copies of three methods
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "methods")); os.makedirs(os.path.join(tmp, "results"))
        base = json.load(open(os.path.join(out_dir, "methods", "AttnGAN.json")))
        cases = {"A": dict(base), "B": dict(base, FID="30.5", RP="50.56"), "C": dict(base, **{"IS*": "40", "CA": "1.82"}),
                 "D": {k: str(float(v) * 1.1) for k, v in base.items()}}
        for name, sc in cases.items():
            json.dump(sc, open(os.path.join(tmp, "methods", name + ".json"), "w"))
        order = os.listdir(os.path.join(tmp, "methods"))
        subprocess.run([sys.executable, os.path.join(REF, "ranking_score.py")], cwd=tmp, env=env, check=True,
                       capture_output=True, text=True)
        saved = open(os.path.join(tmp, "results", "coco_benchmark_results.txt")).read()
    json.dump({"listdir_order": order, "methods": cases, "table": saved},
              open(os.path.join(out_dir, "reference_run_ties.json"), "w"), indent=1)
    print("wrote", out_dir)


if __name__ == "__main__":
    main()
