# Round-6 session 4: hardware counters of the two PageRank kernels on the final code (the only phase-B counters
# were round 3's, before the lane-interleaved streams) -- tools/pmc_pb.sh, twelve counter sets, collection restricted to the two kernels
bash tools/pmc_pb.sh r06_pb 27 2>&1 | tail -20
python3 tools/pmc_pb_summary.py r06_pb; cp profiles/r06_pb_phaseB_counters.md gpurun_out/pmc_r06_pb/ 2>/dev/null; sed -n 1,60p profiles/r06_pb_phaseB_counters.md
