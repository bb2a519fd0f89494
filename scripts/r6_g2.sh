mkdir -p gpurun_out/r6b
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue scripts/micro/valu_issue.hip 2>/dev/null && timeout 300 /tmp/valu_issue > gpurun_out/r6b/valu_issue.md 2> gpurun_out/r6b/valu_issue.err
cd /tmp && export TMPDIR=/tmp
# calibrate the busy counters on kernels whose issue rate is known from the table above
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVES -d /tmp/cal -o cal --output-format csv -- /tmp/valu_issue > /dev/null 2>&1
find /tmp/cal -name "*.csv" | head
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/cal/**/*counter_collection.csv', recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
print(rows[0].keys())
agg = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'], r['Dispatch_Id'])
    agg.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    agg[key]['wg'] = r.get('Workgroup_Size'); agg[key]['grid'] = r.get('Grid_Size')
out = open('/root/repo/gpurun_out/r6b/calib.txt', 'w')
for (k, d), v in agg.items():
    out.write(f"{k} disp {d} wg {v.get('wg')} grid {v.get('grid')} " + " ".join(f"{a}={b:.0f}" for a, b in v.items() if a not in ('wg', 'grid')) + "\n")
PY
cd $GRAFT_REPO_ROOT
cat gpurun_out/r6b/valu_issue.md | tail -40
head -30 gpurun_out/r6b/calib.txt
