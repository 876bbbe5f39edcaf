"""python tests/golden/compare_fixtures.py OLD_DIR [names...]: the committed oracle fixtures (tests/golden/step_*.safetensors) against another
set of the same files -- what a regeneration by make_golden_step.py changed.  Per fixture: the largest relative L2 difference over its
tensors, the largest relative difference over its scalars, and the two source stamps.  (The oracle is deterministic on one host and thread
count; across hosts torch's CPU sqrt / BLAS summation order move last bits)"""
import glob
import json
import os
import sys

from safetensors import safe_open

HERE = os.path.dirname(os.path.abspath(__file__))


def read(path):
    t, meta = {}, {}
    with safe_open(path, framework="pt") as f:
        for k in f.keys():
            t[k] = f.get_tensor(k)
        md = f.metadata() or {}
    return t, json.loads(md.get("scalars", "{}")), json.loads(md["stamp"]) if md.get("stamp") else None


def flat(v):
    if isinstance(v, (list, tuple)):
        return [y for x in v for y in flat(x)]
    return [float(v)] if isinstance(v, (int, float)) else []


def main():
    old_dir, names = sys.argv[1], sys.argv[2:]
    worst = 0.0
    for new_path in sorted(glob.glob(os.path.join(HERE, "step_*.safetensors"))):
        base = os.path.basename(new_path)
        if names and base[5:-12] not in names:
            continue
        old_path = os.path.join(old_dir, base)
        if not os.path.exists(old_path):
            print("%-40s (new fixture)" % base)
            continue
        (tn, sn, stn), (to, so, sto) = read(new_path), read(old_path)
        dt, wk = 0.0, None
        for k in tn:
            if k not in to or tn[k].shape != to[k].shape:
                dt, wk = float("inf"), k
                break
            a, b = tn[k].double(), to[k].double()
            d = float((a - b).norm() / (b.norm() + 1e-300))
            if d > dt:
                dt, wk = d, k
        ds, ws = 0.0, None
        for k in sn:
            a, b = flat(sn[k]), flat(so.get(k, []))
            if len(a) != len(b):
                continue          # (bookkeeping entries such as oracle_seconds may be lists of another length)
            for x, y in zip(a, b):
                if k.endswith("seconds"):
                    continue
                d = abs(x - y) / (abs(y) + 1e-300)
                if d > ds:
                    ds, ws = d, k
        worst = max(worst, dt, ds)
        print("%-44s tensors %d, worst rel-L2 %.2e (%s); scalars worst rel %.2e (%s); stamp %s -> %s"
              % (base, len(tn), dt, wk, ds, ws, (sto or {}).get("oracle_sha256", "none")[:10], (stn or {}).get("oracle_sha256", "none")[:10]))
    print("worst over all: %.2e" % worst)


if __name__ == "__main__":
    main()
