#!/bin/bash
# Round 4, call r: the whole GPU suite on the final host code, the parser's before / after on the box's cores, then the round's
# closing profile set (bench line, kernel stats per leg with the audio launches by grid size, 8 ranks on this one GPU)
set -u
mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "gpu suite: $(( $(date +%s) - t0 )) s"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
{
  python tools/bench_parse.py --threads 1,16 --repeat 6
  python tools/bench_parse.py --root tools/parse_history/pairs_by_vlc_loop --threads 1,16 --repeat 6
} > gpurun_out/r4r_parse_before_after.txt 2>&1
cat gpurun_out/r4r_parse_before_after.txt
bash tools/ab/final_profiles_round4.sh round4_zz quick 2>&1 | tail -40
