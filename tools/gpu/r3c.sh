cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c; mkdir -p $O
bash tools/bench_variants.sh base nostore w5 base > $O/variants.txt 2>&1
cat $O/variants.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "GRBM_GUI_ACTIVE|GRBM_COUNT|SQ_BUSY_CU_CYCLES|SQ_CYCLES|SQ_VALU_BUSY|SQ_INST_CYCLES|SQ_ACTIVE_INST_VALU|SQ_THREAD_CYCLES_VALU|SQ_BUSY_CYCLES" | head -40 > $O/avail.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/clk -- python3 $R/bench.py --no-cpu-baseline --no-variants --steps 3 --warmup 1 > $O/clk_bench.json 2> $O/clk.err
ls -R $O/clk | head -20
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3c'
cc=glob.glob(O+'/clk/**/*counter_collection.csv', recursive=True)
kt=glob.glob(O+'/clk/**/*kernel_trace.csv', recursive=True)
print(cc, kt)
dur={}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']), r['Kernel_Name'])
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(set)
for r in csv.DictReader(open(cc[0])):
    k=r['Kernel_Name'].split('(')[0].replace('void dabhip::(anonymous namespace)::','')
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
    if r['Dispatch_Id'] in dur: acc[k]['_ns_'+r['Dispatch_Id']]=dur[r['Dispatch_Id']][0]
for k,v in acc.items():
    ns=sum(x for kk,x in v.items() if kk.startswith('_ns_'))
    c={kk:x for kk,x in v.items() if not kk.startswith('_ns_')}
    if ns>2e5: print(k, len(n[k]), 'ns', ns, c, 'GUI_ACTIVE/ns', c.get('GRBM_GUI_ACTIVE',0)/max(ns,1))
PY
