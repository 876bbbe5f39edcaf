"""Turn the two PMC passes of tools/pmc_gemm8p.py (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs, rocpd sqlite output)
into the small JSON bench.py reads `roofline.traffic` from.  usage: pmc_traffic_json.py <fetch.db> <write.db> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports HALF the bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section) -- the
eviction kernel of the same trace (402.7 MB read + 402.7 MB written per launch) is reported beside it as the calibration."""
import collections, json, re, sqlite3, sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, disp, dur, c, v in cur.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
        if c != counter:
            continue
        a = acc[re.sub(r"\s+", " ", name)[:100]]
        a[0] += 1; a[1] += v; a[2] += dur
    return {k: dict(launches=n, per_launch_KB=v / n, duration_us=d / n / 1e3) for k, (n, v, d) in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
g = [k for k in fetch if "pcm_gemm8p_kernel" in k][0]
flush = [k for k in fetch if "elementwise" in k or "vectorized" in k]
M, Ci, Co = 131072, 320, 320
alg_read = (M * Ci + M * 64 + Co * 9 * Ci + Co * 64) * 2 / 1e6
alg_write = M * Co * 2 / 1e6
out = {
    "what": "PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs) on %s, conv3x3 M=131072 (32x64x64 px), 320->320 ch + LoRA "
            "segment, operands evicted between launches (tools/pmc_gemm8p.py)" % g,
    "algorithmic_MB": {"read": round(alg_read, 1), "write": round(alg_write, 1)},
    "kernel": {"FETCH_SIZE_KB_per_launch": fetch[g]["per_launch_KB"], "WRITE_SIZE_KB_per_launch": write[g]["per_launch_KB"],
               "launches": fetch[g]["launches"], "duration_us_under_pmc": fetch[g]["duration_us"],
               "traffic_MB_corrected": round((2 * fetch[g]["per_launch_KB"] + write[g]["per_launch_KB"]) * 1024 / 1e6, 1)},
    "calibration_kernels": {k: {"FETCH_SIZE_KB_per_launch": fetch[k]["per_launch_KB"], "WRITE_SIZE_KB_per_launch": write.get(k, {}).get("per_launch_KB")}
                            for k in flush},
}
out["traffic_over_algorithmic"] = round(out["kernel"]["traffic_MB_corrected"] / (alg_read + alg_write), 2)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
