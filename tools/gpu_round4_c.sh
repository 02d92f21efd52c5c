#!/bin/bash
# round 4, third GPU call: the device-packed stage with the copy stream and the packer's LDS window
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r4c_build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_device_pack.py tests/test_gpu_mixed.py tests/test_gpu_sparse.py -x -q > gpurun_out/r4c_pytest.log 2>&1
echo "tests rc=$?"
tail -5 gpurun_out/r4c_pytest.log
timeout 400 python tools/hostbench/sweep.py > gpurun_out/r4c_sweep.txt 2>&1
echo "sweep rc=$?"
grep "pictures/s" gpurun_out/r4c_sweep.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r4c_prof" -o r4c -- python "$GRAFT_REPO_ROOT/tools/hostbench/sweep.py" quick > "$GRAFT_REPO_ROOT/gpurun_out/r4c_prof_sweep.txt" 2>&1)
echo "prof rc=$?"
python tools/rocpd_stats.py gpurun_out/r4c_prof/r4c_results.db 2>&1 | head -12
timeout 600 python tools/sweep_dense_share.py 256 > gpurun_out/r4c_dense_share_crossover.txt 2>&1
echo "crossover rc=$?"
cat gpurun_out/r4c_dense_share_crossover.txt
