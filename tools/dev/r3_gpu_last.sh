#!/bin/bash
# round-3 closing validation: the whole GPU suite and smoke (what the driver runs at round end)
OUT=gpurun_out/${1:-r3z}
mkdir -p $OUT
(timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
tail -4 $OUT/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log)
