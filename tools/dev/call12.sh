#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_c12
mkdir -p $OUT; cd $ROOT
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mm and not campaign" 2>&1 | tail -4 | tee $OUT/mm_tests.txt
REPS="1 2 3" bash tools/ab.sh gpurun_ab/lib_oldbase.so cur gpurun_ab/lib_widenopf.so 2>&1 | tee $OUT/ab.txt
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee $OUT/clone_time.txt
import torch, coast_amd as ca
eng = ca.Engine(0); eng.set_profiling(True)
g = torch.Generator(device="cuda").manual_seed(1)
f = torch.randint(-2**31, 2**31, (8192, 256, 256), dtype=torch.int32, device="cuda", generator=g)
s = torch.randint(-2**31, 2**31, (8192, 256, 256), dtype=torch.int32, device="cuda", generator=g)
r = torch.empty_like(f)
for name, cfg in (("tmr", ca.XmrConfig(3)), ("tmr_clone", ca.XmrConfig(3, 0, ca.F_CLONE_STAGING)), ("dwc", ca.XmrConfig(2)), ("dwc_clone", ca.XmrConfig(2, 0, ca.F_CLONE_STAGING)), ("none", ca.XmrConfig(1)), ("tmr", ca.XmrConfig(3))):
    for _ in range(3): eng.mm_batch(f, s, out=r, cfg=cfg)
    torch.cuda.synchronize(); eng.reset_stats()
    for _ in range(10): eng.mm_batch(f, s, out=r, cfg=cfg)
    torch.cuda.synchronize(); st = eng.stats()
    print(name, "kernel_ms %.3f" % (st["kernel_ms"] / 10), "errors", st["errors_corrected"], "dwc", st["dwc_detected"])
P
