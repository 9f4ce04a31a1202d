export TMPDIR=/tmp
for cfg in 512,1 512,0; do
 for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=gpurun_out/pmcx/$(echo $cfg$c | tr -c 'A-Za-z0-9' '_')
  EXP_ONLY=$cfg EXP_STREAMS=4096 timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/exp_spectrum.py spectrum > /dev/null 2>&1 || echo fail $cfg $c
  python3 - "$d" "$cfg" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda:[0,0])
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spectrum' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,(v,n) in acc.items(): print(sys.argv[2],k,v/n)
PY
 done
done
