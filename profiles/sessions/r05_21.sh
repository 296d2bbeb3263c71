# Round-5 session 21: the counter passes that did not fit session 2 (address translation, DRAM-side requests, L2 hits, wave waits), on the twelve
# candidates of the plan's own placement search
bash tools/pr_place_pmc.sh gpurun_out/r05s21 > gpurun_out/r05s21.txt 2>&1; tail -120 gpurun_out/r05s21.txt | cut -c1-230
