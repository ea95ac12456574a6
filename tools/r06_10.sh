# round 6: full artefact pass at this commit (bench lines, rocprofv3 stats, PMC, SQ counters) + the whole GPU suite
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/r06j_gpu_tests.txt; cat $O/r06j_gpu_tests.txt
bash tools/profile_round.sh r06j 2>&1 | tail -30
