#!/bin/bash
# round 4, first GPU call: the device-packed stage — tests, hand-over sweep, default bench line, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r4a_build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_device_pack.py -x -q > gpurun_out/r4a_pytest_device_pack.log 2>&1
echo "device_pack tests rc=$?"
tail -5 gpurun_out/r4a_pytest_device_pack.log
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_device_pack.py > gpurun_out/r4a_pytest_gpu.log 2>&1
echo "gpu suite rc=$?"
tail -5 gpurun_out/r4a_pytest_gpu.log
timeout 400 python tools/hostbench/sweep.py > gpurun_out/r4a_sweep.txt 2>&1
echo "sweep rc=$?"
grep "pictures/s" gpurun_out/r4a_sweep.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
echo "bench rc=$?"
tail -3 gpurun_out/r4a_bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r4a_prof" -o r4a -- python "$GRAFT_REPO_ROOT/tools/hostbench/sweep.py" quick > "$GRAFT_REPO_ROOT/gpurun_out/r4a_prof_sweep.txt" 2>&1)
echo "prof rc=$?"
find gpurun_out/r4a_prof -name "*stats*" | head
