# Round-5 session 50: PageRank's placement spread -- allocation or address?  Every candidate of `vals` timed at eight offsets inside itself
# (tools/pr_place_offsets.py), two fresh processes
mkdir -p gpurun_out
for i in 1 2; do
  timeout 900 python3 tools/pr_place_offsets.py 27 10 > gpurun_out/r05s50_run$i.out 2> gpurun_out/r05s50_run$i.txt
  grep "place\]" gpurun_out/r05s50_run$i.txt | grep -v "+" | head -14 | cut -c1-150
done
