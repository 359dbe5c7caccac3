bash tools/gpu_r06.sh suite r06X
grep -n "FAILED\|ERROR" gpurun_out/r06X_pytest.txt | head -5
bash tools/gpu_r06.sh evidence r06
echo 6371535 > gpurun_out/r06_evidence_commit.txt
