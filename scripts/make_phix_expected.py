"""Writes tests/golden/phix_expected.json: what the CPU oracle returns for the reference's PhiX fixture (tests/golden/*.gz, copied
from the reference's resources/), read by read -- in six runs (sample1 / sample2 single-ended and the pairs, each with keys placed
from the qualities and as for quality-less input): "runs" = the top site of every read when the flow stops after the rescue stage
(finalStage = 0), "final" = the record BBMap prints after the final alignment stage (mapped, strand, start, stop, mapScore, paired,
ambiguous, match string), and the score of the fill against each read's TRUTH window (the coordinates in its name +- SLOW_ALIGN_PADDING).  tests/test_golden_phix.py asserts the oracle (and, on the GPU,
the device mapper) against this table field by field, so that a change that moves a single read shows.  The table is the
restatement's output pinned at the commit that wrote it -- not output of the reference (no JVM in this image).
Run from the repository root: python scripts/make_phix_expected.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden_phix import fixture_runs, fixture_runs_pacbio, truth_window_scores, truth_window_scores_pacbio      # noqa: E402

if __name__ == "__main__":
    table = {"runs": {}, "final": {}, "truth_window": {}}
    for name, run in fixture_runs().items():
        fin = run["oracle"]()
        f = fin["final"]
        table["final"][name] = [[int(f["mapped"][i]), int(f["strand"][i]), int(f["start"][i]), int(f["stop"][i]), int(f["mapScore"][i]),
                                 int(f["paired"][i]), int(f["ambiguous"][i]), fin["fmatch"][i][:f["match_len"][i]].tobytes().decode()]
                                for i in range(len(f))]
        out = run["oracle"](final_stage=0)
        top = out["sites"][:, 0]
        table["runs"][name] = [[int(out["nsites"][i]), int(top["strand"][i]), int(top["start"][i]), int(top["stop"][i]), int(top["slowScore"][i])]
                               if out["nsites"][i] > 0 else [int(out["nsites"][i]), 0, 0, 0, 0] for i in range(len(top))]
    for which in (1, 2):
        table["truth_window"]["sample%d" % which] = truth_window_scores(which)
    # the same reads through mapPacBio's classes (BBIndexPacBio, BBMapThreadPacBio, MultiStateAligner9PacBio): top sites after scoreSlow
    table["pacbio_runs"], table["pacbio_truth_window"] = {}, {}
    for name, run in fixture_runs_pacbio().items():
        out = run["oracle"]()
        top = out["sites"][:, 0]
        table["pacbio_runs"][name] = [[int(out["nsites"][i]), int(top["strand"][i]), int(top["start"][i]), int(top["stop"][i]), int(top["slowScore"][i])]
                                      if out["nsites"][i] > 0 else [int(out["nsites"][i]), 0, 0, 0, 0] for i in range(len(top))]
    for which in (1, 2):
        table["pacbio_truth_window"]["sample%d" % which] = truth_window_scores_pacbio(which)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "phix_expected.json")
    with open(path, "w") as f:
        json.dump(table, f, separators=(",", ":"))
    print("wrote", path, {k: len(v) for k, v in table["runs"].items()})
