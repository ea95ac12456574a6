# round 6: final artefact pass (tag r06r) + the whole GPU suite at the hbwd16 / entry commit
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/r06r_gpu_tests.txt; cat $O/r06r_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/r06r_smoke.txt
bash tools/profile_round.sh r06r 2>&1 | tail -12
