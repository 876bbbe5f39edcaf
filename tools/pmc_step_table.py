"""One table of the north star's evidence (MFMA utilisation, HBM GB/s) per kernel of ONE eager bs-16 step, from three separate rocprofv3
--pmc passes over `bench.py --steps 1 --warmup 0 --no-graph` (tools/jobs/r05_e_pmc.sh):
  pass S: SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
  pass F: FETCH_SIZE        pass W: WRITE_SIZE        (TCC slots: the two do not fit one pass, MI355X_MICROARCH.md)
Columns
  mfma_util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): busy cycles summed over the chip's 1024 SIMDs against active cycles summed over
               the 8 XCD instances of GRBM.  Calibration: the dominant GEMM's value reproduces its flop-derived fraction of the 2.5 PFLOP/s MFMA peak
               (bench.py roofline.frac) and the attention forward's equals 14 MFMAs x 32 cycles over its cycles per key tile.
  valu/mfma  = non-MFMA VALU instructions per MFMA instruction;  wait_any = share of wave cycles parked on s_waitcnt / barriers
  rd/wr MB   = FETCH_SIZE x 2 (gfx950 reports half the bytes of wide coalesced reads, same guide) / WRITE_SIZE, per launch, fabric side (Infinity-Cache
               hits included);  GB/s = (rd + wr) / average duration of the S pass
usage: pmc_step_table.py <S.db> <F.db> <W.db> [rows]"""
import collections
import re
import sqlite3
import sys


def key(name):
    return re.sub(r"\s+", " ", re.sub(r"\(.*", "", name))[-58:]


def load(db):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(float)
    for name, d_id, d, c, v in sqlite3.connect(db).cursor().execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events"):
        k = key(name)
        agg[k][c] += v
        if d_id not in disp[k]:
            disp[k].add(d_id); dur[k] += d
    return agg, {k: len(v) for k, v in disp.items()}, dur


S, nS, durS = load(sys.argv[1])
F, nF, _ = load(sys.argv[2])
Wr, nW, _ = load(sys.argv[3])
rows = sorted(S, key=lambda k: -durS[k])[: int(sys.argv[4]) if len(sys.argv) > 4 else 32]
# the table names the kernel sources it was measured on (round 6): bench.py copies the dominant kernel's row into its JSON line only next to
# this stamp and marks it STALE when the stamp is not the id of the library it runs (pcm_amd/build.py source_id)
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "phased-consistency-model_amd"))
try:
    from pcm_amd import build as _build
    print("# source_id: %s" % _build.source_id())
except Exception as e:      # (a table without a stamp is reported as unstamped by bench.py)
    print("# source_id unavailable: %s" % e)
print("%-58s %5s %8s %9s %9s %8s %9s %9s %8s" % ("kernel", "calls", "dur_ms", "mfma_util", "valu/mfma", "wait_any", "rd_MB/l", "wr_MB/l", "GB/s"))
tot = 0.0
for k in rows:
    a = S[k]
    gui = max(a.get("GRBM_GUI_ACTIVE", 0.0), 1.0) * 128.0
    wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    im, iv = a.get("SQ_INSTS_MFMA", 0.0), a.get("SQ_INSTS_VALU", 0.0)
    rd = 2.0 * F.get(k, {}).get("FETCH_SIZE", 0.0) / max(nF.get(k, 1), 1) / 1e3          # KB -> MB, x2
    wr = Wr.get(k, {}).get("WRITE_SIZE", 0.0) / max(nW.get(k, 1), 1) / 1e3
    avg_us = durS[k] / nS[k] / 1e3
    tot += durS[k] / 1e6
    print("%-58s %5d %8.2f %9.3f %9s %8.3f %9.1f %9.1f %8.0f" % (k, nS[k], durS[k] / 1e6, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / gui,
          ("%.2f" % ((iv - im) / im)) if im else "-", a.get("SQ_WAIT_ANY", 0.0) / wc, rd, wr, (rd + wr) / avg_us * 1e3 if avg_us > 0 else 0.0))
print("# %d kernels listed, %.1f ms of %.1f ms total kernel time in the S pass" % (len(rows), tot, sum(durS.values()) / 1e6))
