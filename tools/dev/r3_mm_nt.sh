#!/bin/bash
# round 3: the mm headline with non-temporal r stores / f loads (COAST_MM_AUX_R / _F), A/B on one box: time, then HBM traffic per launch.
# The three libraries are built beforehand (they travel with the snapshot; gpurun_ab/ is git-ignored):
#   H=$(python -c "from coast_amd import build as b; print(b.source_hash())")
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCOAST_SOURCE_HASH="\"$H\"" -DCOAST_MM_AUX_R=<0|2> -DCOAST_MM_AUX_F=<0|2> \
#         -o gpurun_ab/lib_<nt0|ntR|ntRF>.so coast_amd/csrc/coast_hip.hip
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3p
mkdir -p $OUT
REPS="1 2" STEPS=20 bash tools/ab.sh gpurun_ab/lib_nt0.so gpurun_ab/lib_ntR.so gpurun_ab/lib_ntRF.so 2>&1 | tee $OUT/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in nt0 ntR ntRF; do
  export COAST_LIB_OVERRIDE=$ROOT/gpurun_ab/lib_$v.so
  BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$v -o bench -- $BENCH > /dev/null 2> $OUT/pmc_fetch_$v.err
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$v -o bench -- $BENCH > /dev/null 2> $OUT/pmc_write_$v.err
  python - $OUT $v <<'PY'
import sys,sqlite3,glob
out,v=sys.argv[1],sys.argv[2]
def pmc(kind,name):
    vals=[]
    for db in glob.glob("%s/pmc_%s_%s/**/*.db"%(out,kind,v),recursive=True):
        cur=sqlite3.connect(db).cursor()
        for k,c,val,d in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
            if "mm_mfma_blk2" in k and c==name: vals.append((val,d))
    return sum(x for x,_ in vals)/len(vals), sum(d for _,d in vals)/len(vals)/1e3
try:
    (f,df),(w,dw)=pmc("fetch","FETCH_SIZE"),pmc("write","WRITE_SIZE")
    print(v,"FETCH_SIZE %.0f WRITE_SIZE %.0f -> read %.3f GB written %.3f GB total %.3f GB (algorithmic 12.885) kernel %.0f us"%(f,w,f*2048e-9,w*1024e-9,f*2048e-9+w*1024e-9,df))
except Exception as e:
    print(v,"pmc parse failed",repr(e))
PY
done 2>&1 | tee -a $OUT/ab.txt
find $OUT -name "*.db" -delete
