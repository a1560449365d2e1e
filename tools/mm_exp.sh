# development: build experiment variants of the library (-DMM_EXP=k knobs in mm_mfma_kernel.hip) and time them
# usage (here, before gpurun): bash tools/mm_exp.sh build "0 1 2"      (on the GPU box): bash tools/mm_exp.sh run "0 1 2 base"
# "snap NAME" copies the current library to lib/exp_NAME.so as an A/B baseline
R=${GRAFT_REPO_ROOT:-/root/repo}
mode=$1; shift
if [ "$mode" = snap ]; then cp $R/coast_amd/lib/libcoast_hip.so $R/coast_amd/lib/exp_$1.so; exit 0; fi
for pass in 1 2; do
for k in $1; do
  so=$R/coast_amd/lib/exp_$k.so
  if [ "$mode" = build ]; then
    [ $pass = 1 ] && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMM_EXP=$k -o $so $R/coast_amd/csrc/coast_hip.hip 2>/dev/null &
  else
    echo "variant $k: $(PERF_REPS=30 PERF_WARM=10 COAST_MM_ENGINE=mfma COAST_HIP_LIB=$so python $R/tools/perf_kernels.py --only mm 2>&1 | grep mm256 | sed -E 's/.*(rep[0-9]).*"ms": ([0-9.]{6}).*/\1 \2/' | tr '\n' ' ')"
  fi
done
done
wait
