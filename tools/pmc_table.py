"""Per-kernel utilisation table from a rocprofv3 --pmc pass (rocpd sqlite): SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE summed per kernel name over all dispatches and counter instances.
mfma_busy / valu_busy = busy cycles relative to GRBM_GUI_ACTIVE x 32 (uncalibrated absolute scale: compare rows and the two columns);
wait_any / wait_inst = fraction of wave cycles spent waiting on anything / on instruction issue.   usage: pmc_table.py <db> [rows]"""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set); dur = collections.defaultdict(float)
for name, d_id, d, c, v in cur.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
    k = re.sub(r"\s+", " ", re.sub(r"\(.*", "", name))[-64:]
    agg[k][c] += v
    if d_id not in disp[k]:
        disp[k].add(d_id); dur[k] += d
rows = sorted(agg, key=lambda k: -dur[k])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]
print("%-66s %6s %9s %9s %9s %8s %9s %11s %11s %9s" % ("kernel", "calls", "dur_ms", "mfma_busy", "valu_busy", "wait_any", "wait_inst", "insts_valu", "insts_mfma", "valu/mfma"))
for k in rows:
    a = agg[k]; gui = max(a.get("GRBM_GUI_ACTIVE", 0.0), 1.0) * 32; wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    im = a.get("SQ_INSTS_MFMA", 0.0); iv = a.get("SQ_INSTS_VALU", 0.0)
    print("%-66s %6d %9.2f %9.3f %9.3f %8.3f %9.3f %11.3e %11.3e %9s" % (k, len(disp[k]), dur[k] / 1e6, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / gui,
          a.get("SQ_ACTIVE_INST_VALU", 0.0) / gui, a.get("SQ_WAIT_ANY", 0.0) / wc, a.get("SQ_WAIT_INST_ANY", 0.0) / wc, iv, im,
          ("%.2f" % ((iv - im) / im)) if im else "-"))
