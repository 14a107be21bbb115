#!/bin/bash
# round 5, session 15: per-queue occupancy of the overlapped step (kernel trace)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=/root/repo
for m in bf16 fp32; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/r05q_$m -o step -- python $R/tools/profile_step.py --mode $m --steps 8 --warmup 3 --pipeline > $R/$O/r05q_$m.log 2>&1)
t=$(find $O/r05q_$m -name '*kernel_trace.csv' | head -1)
python tools/queue_busy.py "$t" --steps 5 > $O/r05_queue_busy_$m.txt 2>&1
# keep the last steps of the trace for offline analysis (small)
python - "$t" $O/r05_trace_tail_$m.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
sel=rows[marks[-4]:marks[-1]+1]
w=csv.DictWriter(open(sys.argv[2],'w'),fieldnames=['Queue_Id','Kernel_Name','Start_Timestamp','End_Timestamp','Grid_Size_X','Workgroup_Size_X'],extrasaction='ignore')
w.writeheader(); w.writerows(sel)
PY
rm -rf $O/r05q_$m
cat $O/r05_queue_busy_$m.txt
done
