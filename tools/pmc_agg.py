"""Aggregate a rocprofv3 --pmc rocpd sqlite into per-kernel counter sums (small JSON)."""
import sqlite3, sys, json, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for name, disp, d, c, v in cur.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
    k = re.sub(r"\(.*", "", name)[-60:]
    agg[k][c] += v
    if (disp,) not in seen and c == "SQ_WAVE_CYCLES":
        n[k] += 1; dur[k] += d
out = {k: dict(calls=n[k], dur_ms=dur[k] / 1e6, **{c: v for c, v in cs.items()}) for k, cs in agg.items()}
json.dump(out, open(sys.argv[2], "w"), indent=0)
