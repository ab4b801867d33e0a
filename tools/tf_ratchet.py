"""Maintains tests/golden/tf_measured_baseline.json -- the RATCHET under the teacher-forced whole-path parity test.

`test_trainer_teacher_forced_vs_reference` bounds the parameter-movement / gradient error of the BASELINE-size fixtures at 2e-3 ..
1e-2 (a flipped ReLU unit is a legitimate outcome there), which let an 18 x shift of cfg5/default through in round 4.  The ratchet
closes that: every (case, path, update) has its MEASURED values on record, the test fails when a value exceeds `ratio` x the record,
and this tool refuses to record an entry that sits more than `ratio` x above the quietest path of the same (case, update) unless a
`known_flips` item covers it -- an item names the flipped unit, its float64 pre-activation and the probe output that showed it
(tools/parity_pair.py / tools/parity_probe.py).

    python tools/tf_ratchet.py update gpurun_out/<round>/tf_measured.jsonl     # rewrite the entries from a measurement log of the GPU suite
    python tools/tf_ratchet.py check                                          # validate the committed file (also run by the CPU test suite)
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(REPO, "tests", "golden", "tf_measured_baseline.json")
FIELDS = ("move_all", "move_worst", "grad_all", "grad_worst")


def covered(flips, case, path, upd):
    """A known flip of update u covers the later updates of the same (case, path) as well: the trajectories have parted."""
    for f in flips:
        if f["case"] == case and (f["paths"] == "*" or path in f["paths"]) and f["update"] <= upd:
            return f["id"]
    return None


def validate(doc):
    entries, flips, ratio = doc["entries"], doc["known_flips"], float(doc["ratio"])
    problems = []
    groups = {}
    for key, e in entries.items():
        case, path, upd = key.split("/")
        groups.setdefault((case, int(upd)), []).append((path, e))
    for (case, upd), members in groups.items():
        quiet = {f: min(e[f] for _, e in members) for f in FIELDS}
        for path, e in members:
            loud = [f for f in FIELDS if e[f] > ratio * max(quiet[f], doc["floors"][f])]
            cov = covered(flips, case, path, upd)
            if loud and cov is None:
                problems.append(f"{case}/{path}/{upd}: {', '.join(f'{f} {e[f]:.2e} vs {quiet[f]:.2e} on the quietest path' for f in loud)} "
                                f"-- not covered by a known_flips item (run tools/parity_pair.py {case} {path},<quiet path>, name the unit)")
            if e.get("flip") != cov:
                problems.append(f"{case}/{path}/{upd}: entry says flip={e.get('flip')}, known_flips says {cov}")
    for f in flips:
        for k in ("id", "case", "paths", "update", "layer", "unit", "pre_activation", "evidence"):
            if k not in f:
                problems.append(f"known_flips item {f.get('id')}: missing '{k}'")
        if "evidence" in f and not os.path.exists(os.path.join(REPO, f["evidence"])):
            problems.append(f"known_flips item {f.get('id')}: evidence file {f['evidence']} does not exist")
    return problems


def update(log):
    doc = json.load(open(PATH))
    entries = {}
    for line in open(log):
        r = json.loads(line)
        if "move_all" not in r or r.get("path", "").startswith("kink_free"):
            continue
        key = f"{r['case']}/{r['path']}/{r['update']}"
        e = {f: float(f"{r[f]:.3e}") for f in FIELDS}
        cov = covered(doc["known_flips"], r["case"], r["path"], r["update"])
        if cov:
            e["flip"] = cov
        entries[key] = e
    doc["entries"] = dict(sorted(entries.items()))
    doc["source"] = os.path.relpath(os.path.abspath(log), REPO)
    problems = validate(doc)
    for p in problems:
        print("PROBLEM:", p)
    if problems:
        sys.exit(1)
    with open(PATH, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print(f"{len(entries)} entries written to {os.path.relpath(PATH, REPO)}")


if __name__ == "__main__":
    if sys.argv[1] == "update":
        update(sys.argv[2])
    else:
        probs = validate(json.load(open(PATH)))
        for p in probs:
            print("PROBLEM:", p)
        sys.exit(1 if probs else 0)
