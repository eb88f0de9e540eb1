"""Majority vote, launch by launch, over several tuning runs of one shape: vote_schedule.py out.json in1.json in2.json ...
A single `y4_autotune` run decides each launch from a handful of timed launches and a few of its 110 picks are noise (one run
took the 128x64 tile for conv 37 where every other run, on every box, takes 256x256: 70 against 56 us in a step).  The tile
entry of a conv (signed: a negative id = head of a chained run) is the most frequent value over the runs, ties going to the
first file; the stage / residual-block switches are the first file's."""
import collections
import json
import sys

out, files = sys.argv[1], sys.argv[2:]
runs = [json.load(open(f)) for f in files]
base = dict(runs[0])
n = len(base["tiles"])
assert all(len(r["tiles"]) == n and (r["size"], r["classes"], r["batch"], r["dtype"]) == (base["size"], base["classes"], base["batch"], base["dtype"]) for r in runs)
voted, changed = [], []
for i in range(n):
    votes = collections.Counter(r["tiles"][i] for r in runs)
    top = max(votes.values())
    pick = next(r["tiles"][i] for r in runs if votes[r["tiles"][i]] == top)
    voted.append(int(pick))
    if pick != base["tiles"][i]:
        changed.append((i, base["tiles"][i], pick, dict(votes)))
base["tiles"] = voted
base.pop("path", None)
json.dump(base, open(out, "w"))
print(f"{len(changed)} of {n} entries differ from {files[0]}:")
for c in changed:
    print("  conv", c[0], c[1], "->", c[2], c[3])
