# Round-5 session 57: does phase B follow the placement of `vals` as well?  Both phases of every candidate (tools/pr_place_offsets.py 27 12 ab), two processes
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python3 tools/pr_place_offsets.py 27 12 ab > gpurun_out/r05s57_run$i.out 2> gpurun_out/r05s57_run$i.txt
  grep "place\]" gpurun_out/r05s57_run$i.txt | cut -c1-120
done
