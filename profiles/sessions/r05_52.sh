# Round-5 session 52: ... allocation or MOMENT?  The held candidates timed again when the search is over (tools/pr_place_offsets.py)
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python3 tools/pr_place_offsets.py 27 8 > gpurun_out/r05s52_run$i.out 2> gpurun_out/r05s52_run$i.txt
  grep "place\]" gpurun_out/r05s52_run$i.txt | grep -v " + " | cut -c1-150
done
