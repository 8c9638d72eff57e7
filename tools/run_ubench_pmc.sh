R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pu -o u -- $R/tools/mlp_shape_ubench > $R/gpurun_out/r03_ubench_shape.txt 2>&1
python $R/tools/rocpd_stats.py /tmp/pu/u_results.db 12 shape > $R/gpurun_out/r03_ubench_shape_pmc.md
cat $R/gpurun_out/r03_ubench_shape.txt | grep "us per layer"
python - <<'PY'
import sqlite3
cur=sqlite3.connect('/tmp/pu/u_results.db').cursor()
rows=cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
dur=dict(cur.execute("select name, max(end-start) from kernels group by name").fetchall())
d={}
for n,c,v,k in rows: d.setdefault(n,{})[c]=v
for n,c in d.items():
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        t=[v for k,v in dur.items() if k==n]
        gui=c['GRBM_GUI_ACTIVE']; busy=c['SQ_VALU_MFMA_BUSY_CYCLES']
        print(n[:40], 'mfma busy %.3f'%(busy/(1024*gui/8)), 'clock GHz (max-duration launch) %.3f'%(gui/8/t[0]) if t else '', 'wait_any/wave %.3f'%(c.get('SQ_WAIT_ANY',0)/c.get('SQ_WAVE_CYCLES',1)), 'wait_inst/wave %.3f'%(c.get('SQ_WAIT_INST_ANY',0)/c.get('SQ_WAVE_CYCLES',1)))
PY
