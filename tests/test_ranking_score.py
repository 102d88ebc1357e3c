"""Ranking-score table (section 8 f4) against the reference's own output: its committed result table and two
runs of its script captured by tests/golden/make_golden_ranking.py."""
import json
import os

import numpy as np
import pytest

from tise_toolbox_amd import ranking_score as rs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ranking")


def _rows(table_text):
    return sorted(line for line in table_text.splitlines() if line.startswith("| ") and not line.startswith("| Method"))


def test_matches_reference_run_text_exactly():
    ref = json.load(open(os.path.join(GOLD, "reference_run.json")))
    methods = [f.split(".")[0] for f in ref["listdir_order"] if f.split(".")[1] == "json"]
    _, _, text = rs.compute(os.path.join(GOLD, "methods"), methods)
    assert text == ref["table"]


def test_matches_reference_committed_result_table():
    """The table shipped in the reference repository (other row order: the author's directory order)."""
    want = open(os.path.join(GOLD, "coco_benchmark_results_reference.txt")).read()
    order = [line.split("|")[1].strip() for line in want.splitlines() if line.startswith("| ") and "Method" not in line]
    _, _, text = rs.compute(os.path.join(GOLD, "methods"), order)
    assert text == want
    _, _, any_order = rs.compute(os.path.join(GOLD, "methods"))
    assert _rows(any_order) == _rows(want)            # RS of the published benchmark does not depend on row order


def test_ties_and_small_set(tmp_path):
    ref = json.load(open(os.path.join(GOLD, "reference_run_ties.json")))
    mdir = tmp_path / "methods"
    mdir.mkdir()
    for name, sc in ref["methods"].items():
        json.dump(sc, open(mdir / f"{name}.json", "w"))
    methods = [f.split(".")[0] for f in ref["listdir_order"]]
    _, table, text = rs.compute(str(mdir), methods)
    assert text == ref["table"]
    assert table.shape == (4, 10)


def test_rank_definition():
    scores = {"a": [1, 9, 1, 1, 1, 1, 9, 9, 1], "b": [2, 5, 2, 2, 2, 2, 5, 5, 2], "c": [3, 1, 3, 3, 3, 3, 1, 1, 3]}
    names, table = rs.ranking_table(scores)
    assert names == ["a", "b", "c"]
    np.testing.assert_array_equal(table[:, -1], [6.0, 12.0, 18.0])      # worst = 1 per aspect, best = n per aspect


def test_collect_from_result_files(tmp_path):
    (tmp_path / "fid.txt").write_text("FID: 12.5")
    (tmp_path / "is.txt").write_text("[Inception Score] mean: 30.12345 std: 0.54321")
    (tmp_path / "ois.txt").write_text("O-IS: 4.5 +-  0.1")
    (tmp_path / "ofid.txt").write_text("O-FID: 20.25")
    path = rs.collect("mine", {"FID": str(tmp_path / "fid.txt"), "IS*": str(tmp_path / "is.txt"),
                               "O-IS": str(tmp_path / "ois.txt"), "O-FID": str(tmp_path / "ofid.txt")},
                      {"RP": "50", "SOA-C": "40", "SOA-I": "41", "CA": "2.0", "PA": "45"}, str(tmp_path / "methods"))
    got = json.load(open(path))
    assert got == {"IS*": 30.12345, "FID": 12.5, "RP": 50.0, "SOA-C": 40.0, "SOA-I": 41.0, "O-IS": 4.5, "O-FID": 20.25,
                   "CA": 2.0, "PA": 45.0}
    with pytest.raises(ValueError):
        rs.collect("bad", {}, {}, str(tmp_path / "methods"))


def test_cli_default_behaviour(tmp_path, monkeypatch, capsys):
    import shutil
    shutil.copytree(os.path.join(GOLD, "methods"), tmp_path / "methods")
    (tmp_path / "results").mkdir()
    monkeypatch.chdir(tmp_path)
    rs.main([])
    printed = capsys.readouterr().out
    saved = open(tmp_path / "results" / "coco_benchmark_results.txt").read()
    assert printed.rstrip("\n") == saved
    want = open(os.path.join(GOLD, "coco_benchmark_results_reference.txt")).read()
    assert _rows(saved) == _rows(want)
