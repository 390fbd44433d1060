# kernel timeline of a few bench steps: gaps between consecutive kernels on the GPU (rocprofv3 --kernel-trace)
cfg=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/gaps_$cfg; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O -o r -- python3 bench.py --config $cfg --steps 4 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import sqlite3,glob,re
db=glob.glob("$O/*.db")[0]
c=sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if "kernel_dispatch" in t][0]; ks=[t for t in tabs if "info_kernel_symbol" in t][0]
rows=list(c.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t0=rows[0][0]; prev=None; LAST=(rows[-1][1]-t0)/1e6-float("${2:-15.5}")
for st,en,name in rows:
    gap=(st-prev)/1e6 if prev else 0
    if (st-t0)/1e6 > LAST:
        print("%10.3f ms  gap %8.3f dur %8.3f %s"%((st-t0)/1e6,gap,(en-st)/1e6,re.sub(r".*figh\d*","",name.split("(")[0])[-46:]))
    prev=en
PY
grep '^{' $O/log.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
