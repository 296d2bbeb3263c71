# Round-5 session 20: counter traffic again on the final code: the PageRank iteration (tools/profile_r05.sh with its PMC passes) and BFS / SSSP U[1,255]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PMC=1 timeout 3000 bash tools/profile_r05.sh > gpurun_out/r05_profile.log 2>&1; tail -3 gpurun_out/r05_profile.log
for w in "bfs 27" "sssp_u255 24"; do set -- $w; sed -i 's/^    rocprofv3 --pmc/    timeout 900 rocprofv3 --pmc/' tools/traffic.sh; timeout 2400 bash tools/traffic.sh r05t $1 $2 2>&1 | tail -1; done
ls gpurun_out/r05/fetch/*/ gpurun_out/r05t/bfs | head
