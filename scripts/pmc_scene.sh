cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  d=/tmp/ps_$(echo $grp | cut -c1-5)
  rm -rf $d
  NP=16 timeout 600 rocprofv3 --pmc $grp --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/scripts/scene_batch_trace.py > /tmp/ps.out 2>/tmp/ps.err || tail -3 /tmp/ps.err
  python3 - <<PY
import csv,glob,collections
f=glob.glob("$d/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0].replace("void cvo_dev::","")[:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]==list(acc[k].keys())[0]: cnt[k]+=1
for k,v in acc.items():
    if "k_" in k: print(k.ljust(42), cnt[k], {c: round(x/max(cnt[k],1),1) for c,x in v.items()})
PY
done
tail -1 /tmp/ps.out
