#!/bin/bash
# round-2 GPU job B: current SD1.5 bs-16 picture (per-shape GEMM table, kernel trace), tr-read lane mapping
cd $GRAFT_REPO_ROOT; O=gpurun_out/b; mkdir -p $O; export TMPDIR=/tmp
tools/probes/trread > $O/trread.txt 2>&1
PCM_GEMM_TABLE=$O/gemm_table.txt timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1); echo "prof rc=$?" >> $O/rc.log
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python tools/prof_summary.py $DB 60 > $O/kernel_stats.txt 2>&1
python - "$DB" > $O/schema.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
cur = c.execute("select * from kernels limit 1"); print([d[0] for d in cur.description]); print(cur.fetchone())
PY
python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1
cat $O/rc.log
