#!/bin/bash
# per-kernel time of the factorisation (rocprofv3 kernel stats + the engine's factor trace), with and without look-ahead
O=gpurun_out/panel; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in la nola; do
if [ $v = nola ]; then export FEASTHIP_LOOKAHEAD=0; fi
rm -rf $O/$v
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o p -- python3 tools/factor_trace.py > $O/$v.out 2> $O/${v}_trace.txt
echo == $v; f=$(find $O/$v -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), round(int(r['TotalDurationNs'])/1e6,2), 'ms  avg', round(float(r['AverageNs'])/1e3,1), 'us')
PY
grep "depth  [0-3]" $O/${v}_trace.txt | tail -4
done
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/panel/nola/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_panel_reg<512' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('panel<512,8> durations (us), last 110 launches:', [round(x) for x in d[-110:]])
PY
